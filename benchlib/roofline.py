"""The roofline objects of the JSON line: the dominant kernel against HBM (north_star's metric) and the path against VALU issue cycles
(the resource that binds it)."""
from __future__ import annotations

from .common import HBM_PEAK_GBS, _profile_json, algorithmic_bytes, pmc_traffic

def issue_roofline(W, H, nrefs, ms_frame, prof, held_clock_ghz=None):
    """VALU issue CYCLES per frame against the chip's capacity (256 CUs x 4 SIMDs x 2.4 GHz SIMD-cycles per second).
    profiles/pmc_valu.json: wave64 instructions per launch by opcode class from the committed rocprofv3 --pmc pass and the
    disassembly, and the measured issue cost of each class (scripts/ubench/valu_rates.hip).  Two peaks are quoted: the
    guide's 2 cycles per wave64 VALU instruction (1 229 wave-instr/ns chip-wide) and what this instruction mix can reach
    at its measured per-opcode costs."""
    t = _profile_json("pmc_valu.json")
    g = (t or {}).get(f"{W}x{H}")
    if not g or "cycle_model" not in (t or {}):
        return None
    cm = t["cycle_model"]
    simd_cycles_per_ns = cm["simds"] * cm["clock_ghz"]
    path_keys = [k for k in g if k not in ("_meta", "loop_filter4")]     # (batches launch the loop filter's form 3; form 4 is the one-video kernel)
    insts = {k: (g[k]["per_ref"] * nrefs if "per_ref" in g[k] else g[k]["fixed"]) for k in path_keys}
    cycles = {k: (g[k].get("cycles_per_ref", 0) * nrefs if "per_ref" in g[k] else g[k].get("cycles_fixed", 0)) for k in path_keys}
    tot_i, tot_c = sum(insts.values()), sum(cycles.values())
    ns = ms_frame * 1e6
    out = {"bound": "valu_issue", "unit": "SIMD issue cycles", "peak_simd_cycles_per_ns": simd_cycles_per_ns,
           "path": {"instructions_per_frame": int(tot_i), "issue_cycles_per_frame": int(tot_c),
                    "frac_of_issue_cycles": round(tot_c / (ns * simd_cycles_per_ns), 4),
                    "wave_instr_per_ns": round(tot_i / ns, 1), "frac_of_2cycle_peak": round(tot_i * 2 / (ns * simd_cycles_per_ns), 4),
                    "shader_clock_held_ghz": None if not held_clock_ghz else round(held_clock_ghz, 3),
                    "frac_of_issue_cycles_at_held_clock": None if not held_clock_ghz else round(tot_c / (ns * cm["simds"] * held_clock_ghz), 4)},
           "source": t.get("source"), "cost_source": cm.get("source"), "kernels": {}}
    for k in ("search2", "search1_l0", "mb"):
        if k in prof and prof[k][1] and k in insts:
            kns = prof[k][0] / prof[k][1] * 1e6
            out["kernels"][k] = {"instructions_per_launch": int(insts[k]), "issue_cycles_per_launch": int(cycles[k]),
                                 "avg_launch_ms": round(kns * 1e-6, 5), "frac_of_issue_cycles": round(cycles[k] / (kns * simd_cycles_per_ns), 4),
                                 "frac_of_2cycle_peak": round(insts[k] * 2 / (kns * simd_cycles_per_ns), 4),
                                 "note": "launch time measured with all chunks in flight: other chunks' waves share the SIMDs"}
    return out

