#!/usr/bin/env python3
"""scripts/drop_in_bench.py [--width W --height H --frames N --repeats R --out file] [-- the reference's options]: benchlib/drop_in.py's
measurement on its own (the reference's main() on libvp8hip.so, .y4m -> .ivf on tmpfs, against scripts/native/y4m_to_ivf.cpp)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchlib.drop_in import measure  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--out", default=None)
    a, rest = ap.parse_known_args()
    res = measure(a.width, a.height, a.frames, a.repeats, opts=tuple(rest))
    text = json.dumps(res, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(text + "\n")
    print(text)
