#!/usr/bin/env python3
"""Randomised end-to-end parity on the GPU box: random geometry, quantizer range, SSIM target, GOP/altref periods,
partitions, content (synthetic motion, noise, scene cuts); the native frame loop's frames and filtered
reconstructions against the oracle loop + reference encode_header.
    python scripts/fuzz_parity.py [--cases 40 --seed 1]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from bitstream_cases import expected_frame
from oracle_lib import Oracle
from vp8oclenc_amd import api
from vp8oclenc_amd.driver import InterPathDriver
from vp8oclenc_amd.synth import SynthSequence

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40); ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0 = time.time(); nframes = 0; nbytes = 0; nkey = 0; nrepl = 0; nredo = 0
for case in range(a.cases):
    W, H = 16 * int(rng.integers(1, 41)), 16 * int(rng.integers(1, 31))
    if W < 32 and H < 32:
        W = 32
    qmin = int(rng.integers(0, 100)); qmax = int(min(127, qmin + rng.integers(0, 60)))
    target = float(rng.choice([-1.0, -1.0, 0.9, 0.93, 0.97]))
    gop = int(rng.choice([3, 5, 150])); alt = int(rng.choice([2, 3, 5]))
    P = int(rng.choice([1, 2, 4, 8])); nfr = int(rng.integers(3, 7))
    kind = rng.choice(["synth", "synth", "noise", "cut"])
    seed = int(rng.integers(1, 10000))
    s = SynthSequence(W, H, seed=seed, noise=int(rng.integers(0, 16)))
    s2 = SynthSequence(W, H, seed=seed + 1)
    nrng = np.random.default_rng(seed)
    nz = [(nrng.integers(0, 256, (H, W)).astype(np.uint8), nrng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8),
           nrng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8)) for _ in range(nfr)] if kind == "noise" else None
    host_bs = int(rng.random() < 0.25); dev_params = int(rng.random() < 0.5)
    drv = api.NativeDriver(W, H, gop_size=gop, altref_range=alt, qi_min=qmin, qi_max=qmax, ssim_target=target, num_partitions=P,
                           check_ssim=1, host_bitstream=host_bs, device_params=dev_params)
    ora = Oracle(W, H, target)
    do = InterPathDriver(ora, W, H, gop_size=gop, altref_range=alt, qi_min=qmin, qi_max=qmax, ssim_target=target)
    tag = f"case {case}: {W}x{H} q{qmin}-{qmax} t{target} gop{gop}/{alt} P{P} {kind} seed{seed} hostbs{host_bs} devp{dev_params}"
    for t in range(nfr):
        y, u, v = nz[t] if kind == "noise" else (s2.frame(t) if (kind == "cut" and t >= nfr // 2) else s.frame(t))
        was_key = drv.encode_frame_host(y, u, v)
        got = drv.get_frame()
        out = do.encode_frame(y, u, v)
        assert was_key == (out is None), f"{tag} frame {t}: key decision"
        exp = expected_frame(W, H, do.last_key if out is None else out, out is None, P)
        assert got == exp, f"{tag} frame {t}: bitstream differs ({len(got)} vs {len(exp)} bytes)"
        for p_, q_ in zip(drv.hip.download_last(), ora.download_last()):
            assert np.array_equal(p_, q_), f"{tag} frame {t}: filtered reconstruction"
        nframes += 1; nbytes += len(got); nkey += was_key
        if out is not None:
            nrepl += int(out["replaced"])
    nredo += drv.stats().redone_as_key
    drv.close(); ora.close()
print(f"fuzz seed {a.seed}: {a.cases} cases, {nframes} frames ({nkey} key, {nredo} recoded as key, {nrepl} macroblocks replaced by intra), "
      f"{nbytes} bytes: all identical; {time.time() - t0:.0f} s")
