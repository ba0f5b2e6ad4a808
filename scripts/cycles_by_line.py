#!/usr/bin/env python3
"""Issue cycles of one kernel by SOURCE LINE: hipcc -gline-tables-only, every VALU instruction priced as in issue_cycles.py and
charged to the line its .loc names.  python scripts/cycles_by_line.py kernels_mb.hip k_mb_b [--top 40] [--loop-trips 4]"""
import argparse, collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import issue_cycles as ic

ap = argparse.ArgumentParser()
ap.add_argument("file"); ap.add_argument("kernel"); ap.add_argument("--top", type=int, default=40)
ap.add_argument("--loop-trips", type=int, default=1, help="trip count of the kernel's largest backward branch")
a = ap.parse_args()
asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-gline-tables-only", "-I", os.path.join(ROOT, "include"),
                      "-x", "hip", "--cuda-device-only", "-S", os.path.join(ic.CSRC, a.file), "-o", "-", "-w"], check=True, capture_output=True, text=True).stdout
lines = asm.split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(a.kernel), l))
end = next(i for i, l in enumerate(lines) if "s_endpgm" in l and i > start)
files = {}
for l in lines[:end]:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
seq, labels, cur = [], {}, ("?", 0)
for l in lines[start + 1:end]:
    t = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    lm = re.match(r"^(\.LBB\w+):", t)
    if lm:
        labels[lm.group(1)] = len(seq)
        continue
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    parts = t.split(";")[0].strip().split(None, 1)
    seq.append((parts[0], parts[1] if len(parts) > 1 else "", cur))
loop = None
if a.loop_trips > 1:
    for i, (mn, ops, _) in enumerate(seq):
        if mn.startswith("s_cbranch") and ops.strip() in labels and labels[ops.strip()] < i and (loop is None or i - labels[ops.strip()] > loop[1] - loop[0]):
            loop = (labels[ops.strip()], i)
cyc, cnt = collections.Counter(), collections.Counter()
for i, (mn, ops, c) in enumerate(seq):
    if mn.startswith("v_"):
        w = a.loop_trips if loop and loop[0] <= i <= loop[1] else 1
        cyc[c] += ic.cost_of(mn, ops)[0] * w
        cnt[c] += w
print("total: %d VALU, %.0f issue cycles" % (sum(cnt.values()), sum(cyc.values())))
src = {}
def text(f, n):
    p = os.path.join(ic.CSRC, f)
    if os.path.exists(p):
        src.setdefault(p, open(p).read().split("\n"))
        return src[p][n - 1].strip()[:110] if 0 < n <= len(src[p]) else ""
    return ""
for (f, n), c in sorted(cyc.items(), key=lambda kv: -kv[1])[:a.top]:
    print("%7.0f %5d  %s:%d  %s" % (c, cnt[(f, n)], f, n, text(f, n)))
