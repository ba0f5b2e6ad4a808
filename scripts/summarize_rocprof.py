#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/...) into the small summaries kept under profiles/.

  python scripts/summarize_rocprof.py --tag r01c --geometry 1920x1088 \
      --stats gpurun_out/r01c_kt1/kt1_kernel_stats.csv:1gop gpurun_out/r01c_kt16/kt16_kernel_stats.csv:16gop \
      --fetch gpurun_out/r01c_pmc_fetch/fetch_counter_collection.csv \
      --write gpurun_out/r01c_pmc_write/write_counter_collection.csv

The PMC passes are separate runs (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).  Both counters are
in KiB.  Per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so
it is doubled before it is compared with a byte count; WRITE_SIZE is exact for streaming stores.  The raw values
are kept next to the corrected sum so the correction stays visible."""
import argparse, collections, csv, json, os, shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = {"k_loop_filter3": "loop_filter", "k_loop_filter4": "loop_filter4", "k_loop_filter2": "loop_filter", "k_search2": "search2", "k_mb": "mb",
         "k_search1": "search1", "k_border": "border", "k_pack": "pack", "k_pyramid": "downsample"}


def short(name):
    base = name.split("(")[0].split("::")[-1]
    if base == "" and "k_search2" in name:
        base = "k_search2"
    for k, v in SHORT.items():
        if k in name:
            return v
    return base or name


def pmc(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--geometry", required=True)
    ap.add_argument("--stats", nargs="*", default=[])
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    for item in a.stats:
        path, label = item.split(":")
        shutil.copy(path, os.path.join(out, f"{a.tag}_kernel_stats_{a.geometry}_{label}.csv"))
    if a.fetch and a.write:
        f, w = pmc(a.fetch), pmc(a.write)
        table = {}
        for k in sorted(set(f) | set(w)):
            if k.startswith("__amd"):
                continue
            # a kernel launched with several grid sizes (1-3 references): report the largest grid = the common case
            grid = max(set(f.get(k, {})) | set(w.get(k, {})))
            fv, wv = f.get(k, {}).get(grid, [0.0]), w.get(k, {}).get(grid, [0.0])
            fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
            table[k] = {"grid_size": grid, "launches_sampled": [len(fv), len(wv)], "FETCH_SIZE_KiB_raw": round(fk, 1),
                        "WRITE_SIZE_KiB": round(wk, 1), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
                        "hbm_bytes_per_launch_uncorrected": int((fk + wk) * 1024)}
        p = os.path.join(out, "pmc_traffic.json")
        doc = json.load(open(p)) if os.path.exists(p) else {}
        doc["source"] = f"profiles/pmc_traffic.json ({a.tag}: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; 2*FETCH+WRITE)"
        doc[a.geometry] = table
        json.dump(doc, open(p, "w"), indent=1, sort_keys=True)
        print(json.dumps(table, indent=1))


if __name__ == "__main__":
    main()
