#!/usr/bin/env python3
"""BASELINE configs[3] and configs[4] (per rank) AT THEIR STATED LENGTHS against the CPU oracle loop, and the tables bench.py's
self-checks look up -- too long for the pytest step (minutes of oracle time), so: a script with two halves and committed digests.

  --oracle   (any machine, no GPU) runs the reference's loop on the CPU oracle (tests/oracle_lib.py: oracle/vp8_oracle.c + the
             reference's own encode_header where oracle/_ref is built) and writes tests/golden/full_length/<name>.json:
             per frame the CRC-32 and length of the finished VP8 frame and the CRC-32 of the filtered reconstruction (Y, U, V).
  --verify   (MI355X) codes the same frames with the product -- the native video loop bench.py times (vp8drv_encode_video_device:
             filter-overlap mode, the next frame started before this one's bytes are taken) AND frame by frame with every
             reconstruction downloaded -- and holds every frame against the committed digests.  Prints a log (keep it under profiles/).

Sequences (the frames are bench.py's: vp8oclenc_amd.synth.bench_frames, eight distinct frames cycled):
  config3_4k          3840x2160, 300 frames, -g 150 (the reference's default: key frames at 0 and 150), LAST+GOLDEN+ALTREF, seed 1
  config5_rank<r>     1920x1080 in a 1920x1088 context, ONE closed GOP of 300 frames, seed 1 + r: what rank r of the 8-GPU run codes
                      (bench.py's config5_literal checks its frames against these)
  chunks_<geometry>   for each of the eight start phases, the reconstruction CRCs after 1..N frames of a chunk that starts with its
                      key frame at that phase: what bench.py's batched legs (the headline, the 4K leg) end on, looked up by
                      (phase, frames coded).  Seed 1 (rank 0's).
Reference: the loop of src/vp8enc.cpp:351-488, -g at src/init.h:1431, the GOP restart of src/intra_part.h:1091."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden", "full_length")
ALTREF_RANGE, ND = 5, 8

import numpy as np  # noqa: E402

SEQUENCES = {"config3_4k": dict(W0=3840, H0=2160, seed=1, frames=300, gop=150, start=0)}
for _r in range(8):
    SEQUENCES[f"config5_rank{_r}"] = dict(W0=1920, H0=1080, seed=1 + _r, frames=300, gop=1 << 30, start=0)
SEQUENCES["selftest"] = dict(W0=180, H0=140, seed=5, frames=14, gop=9, start=0)      # (seconds: what the CPU test suite regenerates in full)
TABLES = {"chunks_1920x1080": dict(W0=1920, H0=1080, seed=1, frames=176), "chunks_3840x2160": dict(W0=3840, H0=2160, seed=1, frames=64),
          "chunks_1280x720_last_only": dict(W0=1280, H0=720, seed=1, frames=96, refs="last"),
          # bench.py's other legs (round 6): the four-pass SSIM ladder, the conformant stream (vp8hip_conformant_stream / vp8o_set_conformant_stream:
          # the format's predictor), and ONE video longer than the chunk table (single_stream: phase 0 only).  A finite GOP needs no table of its
          # own: a chunk n frames past a key frame it coded itself stands where a chunk that STARTED with that key frame stands (closed GOPs).
          "chunks_1920x1080_ssim93": dict(W0=1920, H0=1080, seed=1, frames=96, ssim_target=0.93),
          "chunks_1920x1080_conformant": dict(W0=1920, H0=1080, seed=1, frames=96, conformant=1),
          "chunks_1920x1080_phase0_long": dict(W0=1920, H0=1080, seed=1, frames=280, phases=[0])}


def crc_planes(planes):
    return [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in planes]


def oracle_run(W0, H0, seed, frames, gop, start, refs="all", want_bytes=True, log=None, ssim_target=-1.0, conformant=0):
    """the reference's loop on the CPU oracle: per frame (key?, crc32 of the frame's bytes, length, [crc32 of Y, U, V of the filtered reconstruction])"""
    from bitstream_cases import expected_frame
    from oracle_lib import Oracle
    from vp8oclenc_amd.driver import InterPathDriver
    from vp8oclenc_amd.synth import bench_frames
    W, H, _, padded = bench_frames(W0, H0, seed, ND)
    Oracle.lib().vp8o_set_conformant_stream(int(conformant))      # (a global of the oracle library: set for every run)
    ora = Oracle(W, H, ssim_target)
    do = InterPathDriver(ora, W, H, gop_size=gop, altref_range=ALTREF_RANGE, qi_min=0, qi_max=48, ssim_target=ssim_target, check_ssim=True,
                         ref_mask=3 if refs == "all" else 0)
    rows, t0 = [], time.perf_counter()
    for t in range(frames):
        out = do.encode_frame(*padded[(start + t) % ND])
        key = out is None
        crc = ln = None
        if want_bytes:
            b = expected_frame(W, H, do.last_key if key else out, key, 1, dst=(W0, H0))
            crc, ln = zlib.crc32(b), len(b)
        rows.append([int(key), crc, ln, crc_planes(ora.download_last())])
        if log and (t % 25 == 24 or t == frames - 1):
            log(f"    frame {t + 1}/{frames}  {time.perf_counter() - t0:.0f} s")
    ora.close()
    Oracle.lib().vp8o_set_conformant_stream(0)
    return W, H, rows


def do_oracle(names):
    os.makedirs(GOLDEN, exist_ok=True)
    from oracle_lib import Oracle
    log = lambda m: print(m, flush=True)
    log(f"oracle on {Oracle.lib().vp8o_num_threads()} threads")
    for name in names:
        if name in SEQUENCES:
            c = SEQUENCES[name]
            log(f"{name}: {c}")
            W, H, rows = oracle_run(c["W0"], c["H0"], c["seed"], c["frames"], c["gop"], c["start"], log=log)
            doc = dict(kind="sequence", name=name, source=[c["W0"], c["H0"]], coded=[W, H], seed=c["seed"], frames=c["frames"], gop=min(c["gop"], 1 << 30),
                       altref_range=ALTREF_RANGE, distinct_frames=ND, partitions=1, key=[r[0] for r in rows], frame_crc32=[r[1] for r in rows],
                       frame_len=[r[2] for r in rows], recon_crc32=[r[3] for r in rows])
        else:
            c = TABLES[name]
            log(f"{name}: {c}")
            table = []
            for phase in range(ND):
                if phase not in c.get("phases", range(ND)):
                    table.append([])         # (a phase the table does not hold: nothing is looked up there)
                    continue
                W, H, rows = oracle_run(c["W0"], c["H0"], c["seed"], c["frames"], 1 << 30, phase, refs=c.get("refs", "all"), want_bytes=False,
                                        ssim_target=c.get("ssim_target", -1.0), conformant=c.get("conformant", 0))
                table.append([r[3] for r in rows])
                log(f"    phase {phase} done")
            doc = dict(kind="table", name=name, source=[c["W0"], c["H0"]], coded=[W, H], seed=c["seed"], frames=c["frames"], refs=c.get("refs", "all"),
                       ssim_target=c.get("ssim_target", -1.0), conformant=c.get("conformant", 0), phases=list(c.get("phases", range(ND))),
                       altref_range=ALTREF_RANGE, distinct_frames=ND,
                       what="recon_crc32[phase][n - 1] = CRC-32 of (Y, U, V) of the filtered reconstruction after n frames of a chunk whose key frame is frame `phase` of the cycle",
                       recon_crc32=table)
        doc["made_by"] = "scripts/full_length_oracle.py --oracle (CPU oracle loop: tests/oracle_lib.py, driver.InterPathDriver, bitstream_cases.expected_frame)"
        with open(os.path.join(GOLDEN, name + ".json"), "w") as f:
            json.dump(doc, f, separators=(",", ":"))
        log(f"  -> tests/golden/full_length/{name}.json")


def load(name):
    try:
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            return json.load(f)
    except OSError:
        return None


def do_verify(names):
    from vp8oclenc_amd import api
    from vp8oclenc_amd.synth import bench_frames
    bad = 0
    log = lambda m: print(m, flush=True)
    log(f"full-length runs of the product against committed oracle digests (tests/golden/full_length), HIP runtime {api.load_library().vp8hip_runtime_version()}, "
        f"{api.load_library().vp8hip_hw_queues()} hardware queues")
    for name in names:
        doc = load(name)
        if doc is None:
            log(f"{name}: no digest committed -- skipped")
            continue
        W0, H0 = doc["source"]
        W, H, source, _ = bench_frames(W0, H0, doc["seed"], ND)
        dev = [tuple(api.to_device(p) for p in f) for f in source]
        ptrs = [tuple(p.data_ptr() for p in f) for f in dev]
        src = dict(src_width=W0, src_height=H0) if (W0, H0) != (W, H) else {}
        cfg = dict(altref_range=ALTREF_RANGE, qi_min=0, qi_max=48, ssim_target=-1.0, device_params=1, check_ssim=1, **src)
        if doc["kind"] == "sequence":
            n = doc["frames"]
            # pass 1: the loop bench.py times (one video, filter-overlap mode, frames out, the native video loop)
            d = api.NativeDriver(W, H, gop_size=doc["gop"], overlap_filter=1, **cfg)
            d.hip.reserve_frame_path_dense()
            t0 = time.perf_counter()
            frames, keys = d.encode_video_device(n, ptrs, start=0)
            d.hip.synchronize()
            el = time.perf_counter() - t0
            wrong = [t for t in range(n) if (zlib.crc32(frames[t]), len(frames[t])) != (doc["frame_crc32"][t], doc["frame_len"][t])]
            last_ok = crc_planes(d.hip.download_last()) == doc["recon_crc32"][-1]
            st = d.stats()
            d.close()
            log(f"{name}: {W0}x{H0} x {n} frames, native video loop: {n - len(wrong)}/{n} frames byte-identical to the oracle loop (CRC-32 + length), "
                f"final reconstruction {'identical' if last_ok else 'DIFFERS'}, key frames {st.key_frames} (oracle {sum(doc['key'])}), "
                f"{sum(len(f) for f in frames)} bytes, {n / el:.0f} frames/s")
            bad += len(wrong) + (not last_ok) + (st.key_frames != sum(doc["key"]))
            # pass 2: frame by frame, every frame's bytes and every filtered reconstruction
            d = api.NativeDriver(W, H, gop_size=doc["gop"], **cfg)
            wrong_b, wrong_r = [], []
            for t in range(n):
                key = d.encode_frame_device(*ptrs[t % ND])
                b = d.get_frame()
                key = d.resolve() or key
                if (zlib.crc32(b), len(b)) != (doc["frame_crc32"][t], doc["frame_len"][t]) or int(bool(key)) != doc["key"][t]:
                    wrong_b.append(t)
                if crc_planes(d.hip.download_last()) != doc["recon_crc32"][t]:
                    wrong_r.append(t)
            d.close()
            log(f"{name}: frame by frame: {n - len(wrong_b)}/{n} frames byte-identical, {n - len(wrong_r)}/{n} filtered reconstructions identical"
                + (f"; first differing frame {min(wrong_b + wrong_r)}" if wrong_b or wrong_r else ""))
            bad += len(wrong_b) + len(wrong_r)
        else:
            n, tot, wrong = doc["frames"], 0, 0
            cfg = dict(cfg, ssim_target=doc.get("ssim_target", -1.0), conformant_stream=doc.get("conformant", 0))
            for phase in doc.get("phases", range(ND)):
                d = api.NativeDriver(W, H, gop_size=1 << 30, ref_mask=3 if doc["refs"] == "all" else 0, **cfg)
                for t in range(n):
                    d.encode_frame_device(*ptrs[(phase + t) % ND])
                    d.resolve()
                    tot += 1
                    wrong += crc_planes(d.hip.download_last()) != doc["recon_crc32"][phase][t]
                d.close()
            log(f"{name}: {len(doc.get('phases', range(ND)))} start phases x {n} frames: {tot - wrong}/{tot} filtered reconstructions identical to the oracle loop")
            bad += wrong
        for f in dev:
            for p in f:
                p.free()
    log("ALL IDENTICAL" if not bad else f"{bad} DIFFERENCES")
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--verify", action="store_true")
    ap.add_argument("names", nargs="*", help="default: everything")
    a = ap.parse_args()
    names = a.names or ([n for n in SEQUENCES if n != "selftest"] + list(TABLES))
    for n in names:
        if n not in SEQUENCES and n not in TABLES:
            raise SystemExit(f"unknown: {n}; one of {list(SEQUENCES) + list(TABLES)}")
    if a.oracle:
        do_oracle(names)
    if a.verify:
        sys.exit(do_verify(names))
    if not a.oracle and not a.verify:
        ap.print_help()


if __name__ == "__main__":
    main()
