#!/usr/bin/env python3
"""Golden vectors of the coefficient entropy stage, produced by the REFERENCE's own kernels
(count_probs / num_div_denom / encode_coefficients of src/CPU_kernels.cl compiled for x86 by
oracle/build_ref.sh).  Runs only where /root/reference exists; the fixtures it writes
(tests/golden/entropy/*.npz: inputs + expected outputs, no code) travel with the repo.

    python scripts/gen_golden_entropy.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from entropy_cases import from_inter_path, run_stage, synthetic  # noqa: E402
from oracle_lib import ref_stages  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "entropy")


def save(name, coeffs, parts, nz, mbw, mbh, P, ref):
    r = run_stage(ref, coeffs, parts, nz, mbw, mbh, P)
    d = dict(mbw=mbw, mbh=mbh, P=P, coeffs=coeffs, parts=parts, nz=nz, counts=r["counts"], denom=r["denom"],
             probs=r["probs"], sizes=r["sizes"], third_context=r["third_context"])
    for p in range(P):
        d[f"partition_{p}"] = r["partitions"][p]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "partitions", r["sizes"].tolist())


def main():
    ref = ref_stages()
    if ref is None:
        raise SystemExit("oracle/_ref/libvp8ref.so is not built (needs /root/reference)")
    os.makedirs(OUT, exist_ok=True)
    c, p, n = synthetic(6, 4, 11)
    save("synthetic_6x4_p1", c, p, n, 6, 4, 1, ref)
    c, p, n = synthetic(9, 7, 12, density=0.5, big=0.1)
    save("synthetic_9x7_dense_p4", c, p, n, 9, 7, 4, ref)
    c, p, n = synthetic(10, 9, 13, skip=0.5)
    save("synthetic_10x9_skips_p8", c, p, n, 10, 9, 8, ref)
    c, p, n = from_inter_path(176, 144, 5)
    save("interframe_176x144_p2", c, p, n, 11, 9, 2, ref)


if __name__ == "__main__":
    main()
