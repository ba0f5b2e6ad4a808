#!/usr/bin/env python3
"""Per-frame timeline of a rocprofv3 --kernel-trace CSV of a single-stream run: for the last N loop-filter launches, when each
kernel started and ended relative to the previous loop filter's start."""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    m = re.search(r"(k_[a-z0-9_]+)", n)
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else n[:20], r.get("Stream_Id", r.get("Queue_Id", "")), int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)))
ev.sort()
lf = [i for i, e in enumerate(ev) if e[2].startswith("k_loop_filter")]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
for a, b in list(zip(lf, lf[1:]))[-nshow - 1:-1]:
    t0 = ev[a][0]
    print("---- frame period %.1f us" % ((ev[b][0] - t0) / 1e3))
    for e in ev[a:b + 1]:
        print("  %8.1f .. %8.1f  (%6.1f us)  %-22s q%s grid %d" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2], e[3], e[4]))
