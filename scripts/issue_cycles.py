#!/usr/bin/env python3
"""Issue-CYCLE accounting of the hot kernels from their gfx950 disassembly.

    python scripts/issue_cycles.py [--kernel k_search2] [--list-slow-fast]

Counting VALU instructions hides that on gfx950 only a handful of opcodes issue at the full rate.  Measured on an
MI355X, 8 waves per SIMD on every CU (scripts/ubench/valu_cost.hip -> profiles/valu_cost.json), SIMD cycles per
wave64 instruction:
    2.3   v_add/sub/subrev_u32, v_and/or/xor/not_b32, v_lshrrev_b32, v_ashrrev_i32, v_mov_b32, 16-bit VOP2 add/sub/max/
          lshlrev, v_add/mul/fma/fmac_f32 -- in their 32-bit VOP1/VOP2 encoding with VGPR, inline-constant or literal sources
    4.2   everything else: the same opcodes with an SGPR source, SDWA or DPP; v_lshlrev_b32 (!), min/max, compares,
          v_cndmask, every VOP3 (add3, lshl_add, perm, alignbyte, sad, med3, bfe, mad, mul_lo/hi) and every VOP3P
          (v_pk_*, v_dot2, v_dot4)
    8.2   v_ashr_pk_u8_i32 / v_ashr_pk_i8_i32, v_mad_u16; a 32x32 MFMA is priced at the 8 cycles of vector issue it takes
This script compiles a kernel file to assembly (hipcc -S, device only), walks every kernel's instruction stream and
prices it.  Straight-line kernels (k_search2 is fully unrolled) are exact; for kernels with loops the static mix is
scaled to the dynamic SQ_INSTS_VALU count of the committed PMC pass (profiles/pmc_valu.json) by bench.py.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vp8oclenc_amd", "csrc")

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_add_u16", "v_sub_u16", "v_subrev_u16", "v_max_u16", "v_max_i16", "v_min_u16", "v_min_i16", "v_lshlrev_b16",
        "v_lshrrev_b16", "v_ashrrev_i16", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32"}
DOUBLE = {"v_ashr_pk_u8_i32", "v_ashr_pk_i8_i32", "v_mad_u16", "v_mad_i16"}
C_FAST, C_SLOW, C_DOUBLE = 2.31, 4.18, 8.19     # profiles/valu_cost.json
# a 32x32 MFMA keeps the matrix pipe for 32 cycles but the SIMD's VECTOR issue for 8 of them (MI355X_MICROARCH.md, cycle constants):
# priced as what it takes from the vector stream, in the "double" class (k_search2 issues four per wave)
C_MFMA = 8.0


def cost_of(mn: str, operands: str):
    """(cycles, class, why) of one VALU instruction"""
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if base.startswith("v_mfma_"):
        return C_MFMA, "double", "mfma"
    if base in DOUBLE:
        return C_DOUBLE, "double", ""
    if base in FAST:
        if mn.endswith(("_sdwa", "_dpp", "_e64")):
            return C_SLOW, "slow", "encoding"
        srcs = operands.split(",")[1:]
        if any(re.match(r"\s*(s\d+|s\[\d+:\d+\]|vcc|exec|m0|ttmp)", s) for s in srcs):
            return C_SLOW, "slow", "sgpr-source"
        return C_FAST, "fast", ""
    return C_SLOW, "slow", ""


LOOP_TRIPS = {"k_search1<loop>": 4, "k_search1_pl": 4, "k_search1_pl_b": 4}      # (k_search1_plr_b has two nested loops -- references x sub-blocks --: priced through k_search1_pl)    # kernel name -> trip count of its one backward branch (the four 4x4 sub-blocks)


def short_name(mangled: str) -> str:
    m = re.search(r"(k_[a-z0-9_]+)", mangled)
    name = m.group(1) if m else mangled
    if name == "k_search1":      # template <bool SPLIT>
        name += "<split>" if "ILb1E" in mangled else "<loop>"
    return name


def kernels_of(asm_text: str):
    """yield (mangled name, [(mnemonic, operands)]) for every kernel (functions that end in s_endpgm); the body of a loop
    listed in LOOP_TRIPS is repeated trip-count times, so the stream is the dynamic one"""
    for name, body in _kernels_static(asm_text):
        trips = LOOP_TRIPS.get(short_name(name), 1)
        if trips > 1:
            labels = {ops: i for i, (mn, ops) in enumerate(body) if mn == "<label>"}
            for i, (mn, ops) in enumerate(body):
                if mn.startswith("s_cbranch") and ops.strip() in labels and labels[ops.strip()] < i:
                    j = labels[ops.strip()]
                    body = body[:j] + body[j:i + 1] * trips + body[i + 1:]
                    break
        yield name, [(mn, ops) for mn, ops in body if mn != "<label>"]


def _kernels_static(asm_text: str):
    lines = asm_text.split("\n")
    name, body = None, []
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, body = m.group(1), []
            continue
        t = l.strip()
        lm = re.match(r"^(\.LBB\w+):", t)
        if lm and name is not None:
            body.append(("<label>", lm.group(1)))
            continue
        if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        parts = t.split(None, 1)
        if name is not None:
            body.append((parts[0], parts[1] if len(parts) > 1 else ""))
            if parts[0] == "s_endpgm":
                yield name, body
                name, body = None, []


def compile_asm(src: str) -> str:
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"), "-x", "hip",
           "--cuda-device-only", "-S", os.path.join(CSRC, src), "-o", "-", "-w"]
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def price(body):
    cyc = 0.0
    by = collections.Counter()
    n_valu = 0
    avoidable = collections.Counter()
    for mn, ops in body:
        if not mn.startswith("v_"):
            continue
        c, cls, why = cost_of(mn, ops)
        n_valu += 1
        cyc += c
        by[(re.sub(r"_(e32|e64)$", "", mn), cls)] += 1
        if why == "sgpr-source":
            avoidable[re.sub(r"_e32$", "", mn)] += 1
    return n_valu, cyc, by, avoidable


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", nargs="*", default=["kernels_s2.hip", "kernels_me.hip", "kernels_mb.hip", "kernels_lf3.hip", "kernels_lf4.hip", "kernels_rc.hip"])
    ap.add_argument("--kernel", default=None, help="substring of the kernel name to print the opcode table for")
    ap.add_argument("--json", default=None, help="write {kernel: {valu, cycles, cycles_per_instr}} here")
    ap.add_argument("--opcode-json", default=None, help="write the per-opcode table of every kernel (count x price, by cycles) here")
    a = ap.parse_args()
    out = {}
    tables = {}
    for f in a.files:
        for name, body in kernels_of(compile_asm(f)):
            short = short_name(name)
            n, cyc, by, avoidable = price(body)
            price_of = {"fast": C_FAST, "slow": C_SLOW, "double": C_DOUBLE}
            tables[short] = {"file": f, "valu_instructions": n, "issue_cycles": round(cyc, 1), "cycles_per_instruction": round(cyc / max(n, 1), 3),
                             "by_opcode": [{"opcode": mn, "class": cls, "count": k, "cycles": round(k * (C_MFMA if mn.startswith("v_mfma") else price_of[cls]), 1),
                                            "share_of_cycles": round(k * (C_MFMA if mn.startswith("v_mfma") else price_of[cls]) / max(cyc, 1), 4)}
                                           for (mn, cls), k in sorted(by.items(), key=lambda kv: -kv[1] * (C_MFMA if kv[0][0].startswith("v_mfma") else price_of[kv[0][1]]))]}
            out[short] = {"file": f, "static_valu": n, "static_cycles": round(cyc, 1), "cycles_per_instr": round(cyc / max(n, 1), 3),
                          "fast_opcodes_made_slow_by_an_sgpr_source": int(sum(avoidable.values()))}
            print(f"{short:28s} {f:18s} VALU {n:5d}  cycles {cyc:9.0f}  {cyc / max(n, 1):.2f} cyc/instr   fast-ops slowed by an SGPR source: {sum(avoidable.values())}")
            if a.kernel and a.kernel in short:
                for (mn, cls), k in sorted(by.items(), key=lambda kv: -kv[1] * {"fast": C_FAST, "slow": C_SLOW, "double": C_DOUBLE}[kv[0][1]]):
                    c = {"fast": C_FAST, "slow": C_SLOW, "double": C_DOUBLE}[cls]
                    print(f"      {mn:28s} {cls:6s} x{k:4d} = {k * c:7.0f} cycles")
                if avoidable:
                    print("      slowed by an SGPR source:", dict(avoidable))
    if a.opcode_json:
        with open(a.opcode_json, "w") as fjs:
            json.dump({"what": "static (k_search1<loop>: dynamic, loop x4) VALU instruction stream of every kernel by opcode, priced with the per-class issue "
                               "costs measured on the part (profiles/valu_cost.json): count x price, sorted by cycles",
                       "costs": {"fast": C_FAST, "slow": C_SLOW, "double": C_DOUBLE, "mfma_32x32": C_MFMA}, "kernels": tables}, fjs, indent=1)
    if a.json:
        with open(a.json, "w") as fjs:
            json.dump({"costs": {"fast": C_FAST, "slow": C_SLOW, "double": C_DOUBLE, "source": "profiles/valu_cost.json"}, "kernels": out}, fjs, indent=1)


if __name__ == "__main__":
    sys.exit(main())
