#!/usr/bin/env python3
"""profiles/<tag>_valu_busy_by_counters.json: the issue-cycle model (profiles/pmc_valu.json: PMC instruction counts x per-opcode issue
costs) against the hardware's own counters, kernel by kernel, from the passes of scripts/profile_round4.sh:
    --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE        (busy)    VALU-active SIMD cycles and the launch's duration in cycles
    --pmc SQ_BUSY_CYCLES SQ_WAVES                    (busy2)
    --pmc SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES  (mfma)    what k_search2 put on the matrix pipe
each a run of its own on `bench.py --gops-per-gpu 6 --batch 6 --steps 12 --warmup 4 --no-side-legs --cpu-seconds 0` (launches of six
chunks; counter collection serialises the dispatches: every kernel is seen ALONE on the part).

    python scripts/make_valu_busy.py --tag r04c --geometry 1920x1088
"""
import argparse, collections, csv, json, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        acc[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--geometry", required=True)
    a = ap.parse_args()
    g = os.path.join(ROOT, "gpurun_out")
    busy = table(os.path.join(g, f"{a.tag}_pmc_busy", "busy_counter_collection.csv"))
    busy2 = table(os.path.join(g, f"{a.tag}_pmc_busy2", "busy2_counter_collection.csv"))
    mfma = table(os.path.join(g, f"{a.tag}_pmc_mfma", "mfma_counter_collection.csv"))
    model = json.load(open(os.path.join(ROOT, "profiles", "pmc_valu.json")))[a.geometry]
    chunks = 6
    want = {"k_search2_b": ("search2", 3, "frame and reference"), "k_search1_b<false>": ("search1_l0", 3, "frame and reference (level 0)"),
            "k_mb_b": ("mb", 1, "frame"), "k_loop_filter3_b": ("loop_filter", 1, "frame"), "k_strength_segments_b": ("lf_strength", 1, "frame"),
            "k_pack_b": ("pack", 1, "frame")}
    out = {}
    for short, (key, refs, unit) in want.items():
        # the largest grid of the kernel = all three references, six chunks
        cands = [(k, v) for k, v in busy.items() if short.split("<")[0] in k[0] and (("<false>" in k[0]) == ("<false>" in short) or "search1" not in short)]
        if not cands:
            continue
        (name, grid), v = max(cands, key=lambda kv: kv[0][1])
        act = sum(v["SQ_ACTIVE_INST_VALU"]) / len(v["SQ_ACTIVE_INST_VALU"]) * 4          # quad-cycles -> SIMD cycles
        gui = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"]) / 8                   # summed over the 8 XCDs
        e = {"grid_size": grid, "launches_sampled": len(v["SQ_ACTIVE_INST_VALU"]), "valu_active_simd_cycles_per_launch": int(act),
             "launch_cycles": int(gui), "launch_ms_at_2.4GHz": round(gui / 2.4e6, 4), "valu_busy_alone": round(act / (gui * 1024), 4)}
        m = model.get(key)
        if m:
            per_unit_model = m.get("cycles_per_ref", m.get("cycles_fixed"))
            per_unit = act / (chunks * refs)
            e.update({"per_unit_by_counters": int(per_unit), "per_unit_by_the_issue_cycle_model": int(per_unit_model), "unit": unit,
                      "counters_over_model": round(per_unit / per_unit_model, 3)})
        b2 = busy2.get((name, grid))
        if b2:
            e["SQ_BUSY_CYCLES_per_launch"] = int(sum(b2["SQ_BUSY_CYCLES"]) / len(b2["SQ_BUSY_CYCLES"]))
            e["waves_per_launch"] = int(sum(b2["SQ_WAVES"]) / len(b2["SQ_WAVES"]))
        mf = mfma.get((name, grid))
        if mf and "SQ_INSTS_VALU_MFMA_I8" in mf:
            e["mfma_i8_instructions_per_launch"] = int(sum(mf["SQ_INSTS_VALU_MFMA_I8"]) / len(mf["SQ_INSTS_VALU_MFMA_I8"]))
            if "SQ_VALU_MFMA_BUSY_CYCLES" in mf:
                e["mfma_busy_cycles_per_launch"] = int(sum(mf["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(mf["SQ_VALU_MFMA_BUSY_CYCLES"]))
        out[short] = e
    # the frame at 2.8 references, by counters and by the model
    if all(k in out for k in ("k_search2_b", "k_search1_b<false>", "k_mb_b")):
        refs = 2.8
        s1_other = sum(model[f"search1_l{l}"]["cycles_per_ref"] for l in (1, 2, 3, 4))
        fixed_model = sum(model[k]["cycles_fixed"] for k in ("mb", "loop_filter", "lf_strength", "pack", "downsample") if k in model)
        by_model = refs * (model["search2"]["cycles_per_ref"] + model["search1_l0"]["cycles_per_ref"] + s1_other) + fixed_model
        by_counters = refs * (out["k_search2_b"]["per_unit_by_counters"] + out["k_search1_b<false>"]["per_unit_by_counters"] + s1_other) + \
            sum(out[k]["per_unit_by_counters"] for k in ("k_mb_b", "k_loop_filter3_b", "k_strength_segments_b", "k_pack_b") if k in out) + model["downsample"]["cycles_fixed"]
        frame = {"refs_per_frame": refs, "valu_active_simd_cycles_by_counters": int(by_counters), "issue_cycles_by_the_model": int(by_model),
                 "counters_over_model": round(by_counters / by_model, 3),
                 "note": "search1 levels 1-4, the pyramid and the border taken from the model in both sums (small; their largest-grid launches are not separable by grid size in this pass)"}
    else:
        frame = None
    doc = {"source": f"{a.tag}: scripts/profile_round4.sh (passes busy / busy2 / mfma), post-processed by scripts/make_valu_busy.py",
           "units": "SQ_ACTIVE_INST_VALU counts in units of four cycles (the factor 4 of rocprof's VALUBusy); GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs",
           "what": "launches of six chunks at the largest grid of each kernel (all three references), every kernel alone on the part",
           "kernels": out, "frame": frame}
    p = os.path.join(ROOT, "profiles", f"{a.tag}_valu_busy_by_counters.json")
    json.dump(doc, open(p, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
