#!/usr/bin/env python3
"""Randomised end-to-end parity on the GPU box: random geometry, quantizer range, SSIM target, GOP/altref periods,
partitions, content (synthetic motion, noise, scene cuts), source sizes below the coded size (padded on the device) and the
conformant switch; the native frame loop's frames and filtered reconstructions against the oracle loop + reference
encode_header.
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
import vp8_decode
import vp8_parse
from test_parse_roundtrip import check_frame

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=40); ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--decode", type=int, default=1, help="decode the frames of the conformant cases with the tests' RFC 6386 decoder: exactly the device's reconstruction")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0 = time.time(); nframes = 0; nbytes = 0; nkey = 0; nrepl = 0; nredo = 0; ndecoded = 0; nparsed = 0
for case in range(a.cases):
    W, H = 16 * int(rng.integers(1, 41)), 16 * int(rng.integers(1, 31))
    if W < 32 and H < 32:
        W = 32
    big = rng.random() < 0.06            # now and then one of BASELINE's own geometries: 1080p, 4K (configs[2], configs[3]), 720p
    if big:
        W, H = [(1920, 1088), (3840, 2160), (1280, 720)][int(rng.integers(0, 3))]
    qmin = int(rng.integers(0, 100)); qmax = int(min(127, qmin + rng.integers(0, 60)))
    target = float(rng.choice([-1.0, -1.0, 0.9, 0.93, 0.97]))
    gop = int(rng.choice([3, 5, 150])); alt = int(rng.choice([2, 3, 5]))
    P = int(rng.choice([1, 2, 4, 8])); nfr = int(rng.integers(3, 7))
    overlap = int(rng.random() < 0.4)       # vp8hip_filter_overlap: the loop filter on its own stream, the next frame's side work beside it
    kind = rng.choice(["synth", "synth", "noise", "cut"])
    seed = int(rng.integers(1, 10000))
    s = SynthSequence(W, H, seed=seed, noise=int(rng.integers(0, 16)))
    s2 = SynthSequence(W, H, seed=seed + 1)
    nrng = np.random.default_rng(seed)
    nz = [(nrng.integers(0, 256, (H, W)).astype(np.uint8), nrng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8),
           nrng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8)) for _ in range(nfr)] if kind == "noise" else None
    host_bs = int(rng.random() < 0.25); dev_params = int(rng.random() < 0.5)
    conformant = int(rng.random() < 0.3)          # vp8hip_conformant_stream against the oracle's switch of the same meaning
    sw, sh = W, H
    if dev_params and rng.random() < 0.4:         # a source below the coded size: copy_with_padding on the device
        sw, sh = W - 2 * int(rng.integers(0, 8)), H - 2 * int(rng.integers(0, 8))
    src = dict(src_width=sw, src_height=sh) if (sw, sh) != (W, H) else {}
    drv = api.NativeDriver(W, H, gop_size=gop, altref_range=alt, qi_min=qmin, qi_max=qmax, ssim_target=target, num_partitions=P,
                           check_ssim=1, host_bitstream=host_bs, device_params=dev_params, conformant_stream=conformant, overlap_filter=overlap, **src)
    Oracle.lib().vp8o_set_conformant_stream(conformant)
    ora = Oracle(W, H, target)
    do = InterPathDriver(ora, W, H, gop_size=gop, altref_range=alt, qi_min=qmin, qi_max=qmax, ssim_target=target)
    dec = vp8_decode.Decoder() if (conformant and a.decode and not big) else None      # (the tests' decoder is Python: not at 4K)
    pst = vp8_parse.StreamState()
    tag = f"case {case}: {W}x{H} (source {sw}x{sh}) q{qmin}-{qmax} t{target} gop{gop}/{alt} P{P} {kind} seed{seed} hostbs{host_bs} devp{dev_params} conformant{conformant} overlap{overlap}"
    for t in range(nfr):
        y, u, v = nz[t] if kind == "noise" else (s2.frame(t) if (kind == "cut" and t >= nfr // 2) else s.frame(t))
        if src:      # the driver gets the source rectangle, the oracle loop the same rectangle padded by edge replication
            crop = [np.ascontiguousarray(p[:sh // k, :sw // k]) for p, k in ((y, 1), (u, 2), (v, 2))]
            y, u, v = (np.pad(c, ((0, p.shape[0] - c.shape[0]), (0, p.shape[1] - c.shape[1])), mode="edge") for c, p in zip(crop, (y, u, v)))
            was_key = drv.encode_frame_host(*crop)
        else:
            was_key = drv.encode_frame_host(y, u, v)
        got = drv.get_frame()
        was_key = drv.resolve()      # "inter frame" from encode_frame is provisional until check_SSIM's verdict is in
        out = do.encode_frame(y, u, v)
        assert was_key == (out is None), f"{tag} frame {t}: key decision"
        exp = expected_frame(W, H, do.last_key if out is None else out, out is None, P, dst=(sw, sh) if src else None)
        assert got == exp, f"{tag} frame {t}: bitstream differs ({len(got)} vs {len(exp)} bytes)"
        last = drv.hip.download_last()
        for p_, q_ in zip(last, ora.download_last()):
            assert np.array_equal(p_, q_), f"{tag} frame {t}: filtered reconstruction"
        if dec is None and a.decode and not big:      # every other frame is at least read back: header, modes, vectors, every token
            check_frame(vp8_parse.parse_frame(got, pst), do.last_key if out is None else out, out is None, P, f"{tag} frame {t}")
            nparsed += 1
        if dec is not None:
            f, planes = dec.decode(got)
            assert (f.width, f.height) == (sw, sh) or not f.key, f"{tag} frame {t}: display size"
            for p_, q_ in zip(planes, last):
                assert np.array_equal(p_, q_), f"{tag} frame {t}: a decoder does not arrive at the encoder's reconstruction"
            ndecoded += 1
        nframes += 1; nbytes += len(got); nkey += was_key
        if out is not None:
            nrepl += int(out["replaced"])
    nredo += drv.stats().redone_as_key
    drv.close(); ora.close()
    Oracle.lib().vp8o_set_conformant_stream(0)
print(f"fuzz seed {a.seed}: {a.cases} cases, {nframes} frames ({nkey} key, {nredo} recoded as key, {nrepl} macroblocks replaced by intra), "
      f"{nbytes} bytes: all identical; {nparsed} frames read back by the RFC 6386 parser, {ndecoded} frames of conformant cases decoded to the device's reconstruction; {time.time() - t0:.0f} s")
