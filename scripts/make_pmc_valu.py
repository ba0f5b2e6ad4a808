#!/usr/bin/env python3
"""profiles/pmc_valu.json: wave64 VALU instructions per launch of every kernel of the path (rocprofv3 --pmc SQ_INSTS_VALU,
its own pass, on `bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --no-side-legs --cpu-seconds 0`) priced in SIMD issue
CYCLES: every kernel's dynamic instruction stream is walked by scripts/issue_cycles.py with the per-opcode costs of
profiles/valu_cost.json (scripts/ubench/valu_cost.hip); cycles per launch = PMC instructions x that kernel's cycles per
instruction.  bench.py reads the file for its issue_roofline object.

    python scripts/make_pmc_valu.py --pmc gpurun_out/r02x_pmc_valu/valu_counter_collection.csv --geometry 1920x1088 --tag r02x
"""
import argparse
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import issue_cycles as ic  # noqa: E402

FILES = ["kernels_s2.hip", "kernels_me.hip", "kernels_mb.hip", "kernels_lf3.hip", "kernels_lf4.hip", "kernels_rc.hip"]


def cycles_per_instruction():
    out = {}
    for f in FILES:
        for name, body in ic.kernels_of(ic.compile_asm(f)):
            n, cyc, _, _ = ic.price(body)
            out[ic.short_name(name)] = (n, cyc)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", required=True)
    ap.add_argument("--geometry", required=True)
    ap.add_argument("--tag", required=True)
    a = ap.parse_args()
    W, H = (int(v) for v in a.geometry.split("x"))
    cpi = cycles_per_instruction()
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(a.pmc)):
        if r["Counter_Name"] != "SQ_INSTS_VALU":
            continue
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    table = {}

    def put(key, per_ref, instr, kernel):
        if kernel not in cpi:          # single launches of k_mb / k_pack go through the batched kernels (a batch of one)
            kernel += "_b"
        n, cyc = cpi[kernel]
        e = {"per_ref" if per_ref else "fixed": round(instr, 1), "cycles_per_ref" if per_ref else "cycles_fixed": round(instr * cyc / n, 1),
             "static_stream": {"kernel": kernel, "valu": n, "cycles": round(cyc, 1)}}
        table[key] = e

    per_level = collections.defaultdict(list)   # search1: level -> [(instr per ref, variant)]
    plr = collections.defaultdict(list)         # search1, one workgroup for all references (batches): level -> [(instr per launch, launches)]
    s2_batch = []                               # search2 as batches launch it: [(instr per launch, references, launches)]
    mix = collections.Counter()                 # launches of the run by number of references searched (from the level-4 launches)
    for (name, grid), vals in acc.items():
        v = sum(vals) / len(vals)
        if "k_search1" in name:
            variant = "k_search1<split>" if "<true>" in name or "ILb1E" in name else ("k_search1_pl" if "k_search1_pl" in name else "k_search1<loop>")
            bpw = 12 if variant.endswith("<split>") else 48
            for lvl in range(5):
                nblk = ((W >> lvl) // 8) * ((H >> lvl) // 8)
                wgs = (nblk + bpw - 1) // bpw
                if "k_search1_plr" in name:      # one workgroup searches every enabled reference (grid y = 1): a launch's count covers the frame's references
                    if wgs * 256 == grid:
                        plr[lvl].append((v, len(vals)))
                    continue
                for refs in (1, 2, 3):
                    if wgs * 256 * refs == grid:
                        per_level[lvl].append((v / refs, variant, len(vals)))
                        if lvl == 4:
                            mix[refs] += len(vals)
        elif "k_search2" in name:
            nblk = W * H // 64
            nbx = (nblk + 7) // 8
            for refs in (1, 2, 3):
                if nbx * 256 * refs == grid:
                    per_level["s2"].append((v / refs, "k_search2", len(vals)))
                # what batches launch: a workgroup takes four groups of eight blocks (kernels_s2.hip, search2_groups), and the launch carries the
                # strength scan of the member's new frame (ceil(H / 8) workgroups per reference slice; its instructions are taken out below)
                if "k_search2_bs" in name and ((nbx + 3) // 4 + (H + 7) // 8) * 256 * refs == grid:
                    s2_batch.append((v, refs, len(vals)))
        else:
            for short, key in (("k_mb", "mb"), ("k_loop_filter3", "loop_filter"), ("k_loop_filter4", "loop_filter4"), ("k_pyramid", "downsample"), ("k_pack", "pack"), ("k_border", "border"),
                               ("k_strength_segments", "lf_strength")):
                if short + "(" in name or name.endswith(short) or (short in name and "k_mbhdr" not in name):
                    if key in ("downsample", "border") and key in table and table[key]["fixed"] > v:
                        continue     # several grids (one or two frames' pyramids; borders of one plane set): keep the per-frame one
                    put(key, False, v, "k_mb_p" if "k_mb_p" in name else short)     # (k_mb_p: three kinds of wave in a workgroup; priced with the static stream's average cycles per instruction)
    if s2_batch and "lf_strength" in table:      # the headline's form of the kernel, per reference, without the scan that rides in it
        per = sum((v - table["lf_strength"]["fixed"]) / refs * n for v, refs, n in s2_batch) / sum(n for _, _, n in s2_batch)
        per_level["s2"] = [(per, "k_search2", 10 ** 9)]
    avg_refs = sum(r * n for r, n in mix.items()) / max(sum(mix.values()), 1)
    for lvl, items in plr.items():              # what batches launch: per reference = per launch / the run's references per frame
        if avg_refs > 0:
            per = sum(v * n for v, n in items) / sum(n for _, n in items) / avg_refs
            per_level[lvl] = [(per, "k_search1_pl", 10 ** 9)]      # (priced with the one-reference kernel's cycles per instruction: the same instruction mix)
    for lvl, items in per_level.items():
        tot = sum(n for _, _, n in items)
        variant = max(items, key=lambda it: it[2])[1]
        instr = sum(i * n for i, v2, n in items if v2 == variant) / sum(n for _, v2, n in items if v2 == variant)
        put("search2" if lvl == "s2" else f"search1_l{lvl}", True, instr, variant)
    doc = {"source": f"{a.tag}: rocprofv3 --pmc SQ_INSTS_VALU (own pass) on `bench.py --gops-per-gpu 1 --steps 40 --warmup 10 --no-side-legs "
                     f"--cpu-seconds 0`: wave64 VALU instructions per launch; cycles = instructions x the kernel's issue cycles per instruction "
                     f"(scripts/issue_cycles.py walking the disassembly with the per-opcode costs of profiles/valu_cost.json)",
           "cycle_model": {"simds": 1024, "clock_ghz": 2.4, "costs": {"fast": ic.C_FAST, "slow": ic.C_SLOW, "double": ic.C_DOUBLE},
                           "source": "profiles/valu_cost.json (scripts/ubench/valu_cost.hip, 8 waves per SIMD on every CU); peak = 256 CUs x 4 SIMDs x 2.4 GHz SIMD cycles per second"},
           a.geometry: table}
    p = os.path.join(ROOT, "profiles", "pmc_valu.json")
    json.dump(doc, open(p, "w"), indent=1, sort_keys=True)
    print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
