#!/usr/bin/env python3
"""Coefficient entropy stage at BASELINE geometry: device (vp8hip_count_probs + vp8hip_encode_coefficients) against
the CPU oracle on the same coefficients, byte-exact check included.  usage: python scripts/ent_bench.py [W H P frames]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from entropy_cases import run_stage  # noqa: E402
from oracle_lib import Oracle  # noqa: E402
from vp8oclenc_amd import api  # noqa: E402
from vp8oclenc_amd.synth import SynthSequence  # noqa: E402


def main():
    W, H, P, N = (int(a) for a in (sys.argv[1:5] + ["1920", "1080", "8", "12"][len(sys.argv) - 1:]))
    s = SynthSequence(W, H, seed=1)
    mbw, mbh = s.W // 16, s.H // 16
    hip = api.Vp8Hip(s.W, s.H)
    lastqi, _ = api.quantizer_ladders(0, 48)
    hip.upload_last(*s.frame(0))
    hip.profile_enable(["ent_count", "ent_encode"])
    t_dev, t_cpu, nbytes = [], [], 0
    for t in range(1, N + 1):
        y, u, v = s.frame(t)
        red, sharp = api.loopfilter_strength(y)
        hip.set_segments(api.prepare_segments_data(False, lastqi, 0, red, sharp))
        hip.upload_current(y, u, v)
        hip.inter_transform(0, 0, 0, 0)
        hip.synchronize()
        t0 = time.perf_counter()
        probs, denom0 = hip.count_probs(P)
        parts_dev = hip.encode_coefficients(probs, P)
        t_dev.append(time.perf_counter() - t0)
        r = hip.download_results(recon=False)
        nz = hip.debug(api.DBG_MB_NZ)
        coeffs, mbp = np.ascontiguousarray(r["MB_coeffs"]), np.ascontiguousarray(r["MB_parts"])
        t0 = time.perf_counter()
        exp = run_stage(Oracle.stages(), coeffs, mbp, nz, mbw, mbh, P)
        t_cpu.append(time.perf_counter() - t0)
        assert np.array_equal(probs, exp["probs"])
        for p in range(P):
            assert np.array_equal(parts_dev[p], exp["partitions"][p]), (t, p)
        nbytes += sum(len(x) for x in parts_dev)
        hip.loop_filter()
    pr = hip.profile_read()
    mbs = mbw * mbh
    print(f"{s.W}x{s.H}, {P} partitions, {N} frames: {nbytes / N:.0f} bytes of partitions per frame, all byte-exact")
    print(f"  device: wall {1e3 * np.median(t_dev):.3f} ms/frame incl. read-backs; kernels "
          + ", ".join(f"{k} {1e3 * ms / n:.1f} us" for k, (ms, n) in pr.items()))
    print(f"  oracle (1 thread per call, serial coder): {1e3 * np.median(t_cpu):.2f} ms/frame "
          f"-> {mbs / np.median(t_cpu) / 1e3:.0f} k MB/s vs device {mbs / np.median(t_dev) / 1e6:.2f} M MB/s")
    hip.close()


if __name__ == "__main__":
    main()
