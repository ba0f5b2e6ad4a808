#!/usr/bin/env python3
"""How much of the part's time do the kernels of a rocprofv3 --kernel-trace CSV overlap?  Last `nlf` loop-filter launches."""
import csv, re, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nlf = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ev = []
for r in rows:
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); n = m.group(1) if m else r["Kernel_Name"][:16]
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, int(r.get("Grid_Size", 0) or 0)))
ev.sort()
lf = [e for e in ev if e[2].startswith("k_loop_filter")]
t0, t1 = lf[-min(nlf, len(lf))][0], lf[-1][1]
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print("window %.1f ms, %d kernels" % ((t1 - t0) / 1e6, len(sel)))
pts = []
for e in sel:
    pts.append((e[0], 1)); pts.append((e[1], -1))
pts.sort()
cur = 0; last = t0; hist = collections.Counter()
for t, d in pts:
    hist[cur] += t - last; last = t; cur += d
tot = sum(hist.values())
print("kernels running at once (% of time):", {k: round(100 * v / tot, 1) for k, v in sorted(hist.items())})
busy = collections.Counter(); cnt = collections.Counter()
for e in sel:
    busy[e[2]] += e[1] - e[0]; cnt[e[2]] += 1
for k, v in busy.most_common():
    print("  %-24s %5d launches  avg %7.1f us   %.2f running on average" % (k, cnt[k], v / cnt[k] / 1e3, v / (t1 - t0)))
