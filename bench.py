#!/usr/bin/env python3
"""bench.py -- macroblocks/s of the inter-frame path (ME + DCT + loop filter) on N MI355X.

One "step" = one inter frame through the whole hot path behind the C ABI (vp8hip_set_current_device,
vp8hip_set_segments, vp8hip_inter_transform, vp8hip_loop_filter) on synthetic 1080p YUV420 that is
already resident in HBM.  N > 1: one process per GPU (torch.distributed / RCCL for the barrier and
the max-over-ranks time only); every rank encodes its own GOP chunk -- GOPs are independent units
(SURVEY.md section 8e), so there is no data-path collective and scaling is weak.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- the dominant kernel: algorithmic bytes per launch / hipEvent-measured launch time
  cpu_baseline -- the CPU oracle (oracle/vp8_oracle.c, OpenMP) on a bounded sample of the same workload
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# The HIP runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a
# queue serialise.  The workload is many independent GOP chunks, one stream each: give the runtime 24 queues
# (measured on MI355X: 4 -> 27, 8 -> 32, 16 -> 36, 24 -> 40 M MB/s with 16 chunks in flight; 32 and more are slower).
# Must be set before the first HIP call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)

# ALGORITHMIC bytes per launch of each kernel, as a function of the frame
# geometry: mbs macroblocks, b8 = 4*mbs 8x8 blocks, nrefs enabled references.
def algorithmic_bytes(kernel: str, W: int, H: int, nrefs: float) -> float:
    mbs = (W // 16) * (H // 16)
    b8 = 4 * mbs
    if kernel.startswith("search1_l"):
        lvl = int(kernel[-1])
        blocks = ((W >> lvl) // 8) * ((H >> lvl) // 8)
        return 133.0 * blocks * nrefs          # 64 B cur + 64 B ref + 1 B parent MV + 4 B MV out
    if kernel == "search2":
        return (64 + 64 + 4 + 4 + 4) * b8 * nrefs  # cur + ref + MV in + MV out + cost out
    if kernel == "mb":
        return (384 + 384 + 16 + 8 + 800 + 384 + 20) * mbs  # cur + ref + MVs/ref/parts in; coeffs + recon + ids out
    if kernel == "loop_filter":
        return (384 * 2 + 8) * mbs             # recon read + written in place, mask + segment id
    if kernel == "downsample":
        return 0.0
    return 0.0


def pmc_traffic(kernel: str, W: int, H: int):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json; FETCH_SIZE and
    WRITE_SIZE collected in separate passes and corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot
    run the profiler on itself, so this is the last measured value for the same geometry, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        e = t.get(f"{W}x{H}", {}).get(kernel)
        if e:
            return int(e["hbm_bytes_per_launch"]), t.get("source", "profiles/pmc_traffic.json")
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def pmc_valu(W: int, H: int, nrefs: float):
    """wave64 VALU instructions per launch of every kernel (rocprofv3 --pmc SQ_INSTS_VALU, profiles/pmc_valu.json) and
    the measured chip-wide issue rate of this instruction mix (scripts/ubench/valu_peak.hip), or (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_valu.json")) as f:
            t = json.load(f)
        g = t[f"{W}x{H}"]
        per = {k: (e["per_ref"] * nrefs if "per_ref" in e else e["fixed"]) for k, e in g.items()}
        return per, t
    except (OSError, ValueError, KeyError):
        return None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1920)
    ap.add_argument("--warmup", type=int, default=480)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--distinct-frames", type=int, default=8)
    ap.add_argument("--gops-per-gpu", type=int, default=16, help="independent GOP chunks in flight per GPU (1 = one stream)")
    ap.add_argument("--ssim-target", type=float, default=-1.0, help="SSIM_target (reference default -1 = single LQ pass; 0.93 = the 4-pass path)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--host-params", action="store_true",
                    help="take the per-frame segment data from the host mirror (precomputed, untimed) instead of "
                         "computing loop-filter strength and segment data on the device inside every step")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel with hipEvents (adds overhead)")
    ap.add_argument("--bitstream", action="store_true",
                    help="additionally measure the rate with complete VP8 frames delivered to the host (vp8drv_get_frame)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    from vp8oclenc_amd import api
    from vp8oclenc_amd.synth import SynthSequence

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("VP8_BENCH_FORCE_DIST"):   # the variable exercises the RCCL path with one rank (1-GPU boxes)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    # ---- workload: BASELINE.json configs[2] geometry.  Every rank encodes `gops_per_gpu` independent GOP
    # chunks concurrently, one C-ABI context (= one HIP stream) each: closed GOPs are the unit the path
    # shards by, across GPUs and, on a 256-CU part whose loop filter is a latency-bound
    # wavefront on a handful of CUs, also inside one GPU.
    seq = SynthSequence(args.width, args.height, seed=1 + rank)
    W, H = seq.W, seq.H
    mbs = (W // 16) * (H // 16)
    nd = max(2, args.distinct_frames)
    G = max(1, args.gops_per_gpu)
    host_frames = [seq.frame(t) for t in range(nd)]
    dev_frames = [tuple(torch.from_numpy(p).cuda() for p in f) for f in host_frames]
    lastqi, altrefqi = api.quantizer_ladders(0, 48)        # reference defaults, init.h:1548-1603
    seg_last, seg_alt = [], []
    for y, _, _ in host_frames:                             # host parameter producers, outside the timed path
        red, sharp = api.loopfilter_strength(y)
        seg_last.append(api.prepare_segments_data(False, lastqi, 0, red, sharp))
        seg_alt.append(api.prepare_segments_data(False, altrefqi, 0, red, sharp))
    ref_hist = {"frames": 0, "refs": 0}

    class GopStream:
        """One closed GOP: its own context/stream, frame-type state machine and position in the sequence."""

        def __init__(self, k: int):
            self.enc = api.Vp8Hip(W, H, args.ssim_target, device=local)
            self.gop = api.Gop(gop_size=1 << 30, altref_range=5)   # key frame only at the start of the chunk
            self.t = (k * 3) % nd                                   # chunks start at different frames
            self.gop.next()
            self.gop.key_coded()                                    # key frame: coded by the host (out of scope)
            y, u, v = dev_frames[self.t % nd]
            self.enc.set_last_device(y.data_ptr(), u.data_ptr(), v.data_ptr())
            self.gop.frame_done()
            self.t += 1

        def step(self):
            g = self.gop.next()
            i = self.t % nd
            y, u, v = dev_frames[i]
            self.enc.set_current_device(y.data_ptr(), u.data_ptr(), v.data_ptr())
            if args.host_params:   # segment data precomputed by the host mirror, outside the timed region
                self.enc.set_segments(seg_alt[i] if g.current_is_altref else seg_last[i])
            else:                  # get_loopfilter_strength + prepare_segments_data on the device, inside the step
                self.enc.auto_segments(False, altrefqi if g.current_is_altref else lastqi, 0)
            ug, ua = self.gop.inter_flags()
            self.enc.inter_transform(g.prev_is_golden, g.prev_is_altref, ug, ua)
            self.enc.loop_filter()
            self.gop.frame_done()
            self.t += 1
            ref_hist["frames"] += 1
            ref_hist["refs"] += 1 + ug + ua

    class NativeGopStream:
        """The same, through the native frame loop (vp8_driver.cpp): one C call per frame, parameters on the device."""

        def __init__(self, k: int):
            self.drv = api.NativeDriver(W, H, device=local, gop_size=1 << 30, altref_range=5, qi_min=0, qi_max=48,
                                        ssim_target=args.ssim_target, device_params=1, check_ssim=0)
            self.enc = self.drv.hip
            self.t = (k * 3) % nd
            y, u, v = dev_frames[self.t % nd]
            assert self.drv.encode_frame_device(y.data_ptr(), u.data_ptr(), v.data_ptr())   # frame 0 of the chunk: key
            self.t += 1
            self.ptrs = [tuple(p.data_ptr() for p in f) for f in dev_frames]

        def step(self):
            self.drv.encode_frame_device(*self.ptrs[self.t % nd])
            self.t += 1
            st = self.drv.stats()
            ref_hist["frames"] += 1
            ref_hist["refs"] += 1 + st.last_use_golden + st.last_use_altref

    native = not args.host_params
    streams = [(NativeGopStream if native else GopStream)(k) for k in range(G)]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- warmup, with every kernel timed once to find the dominant one -----------------------------
    nwarm = max(args.warmup, 1)
    streams[0].enc.profile_enable(api.K_NAMES)
    for i in range(nwarm):
        streams[i % G].step()
    torch.cuda.synchronize()
    warm = streams[0].enc.profile_read()
    n0 = len(range(0, nwarm, G))
    per_frame = {k: ms / max(n0, 1) for k, (ms, n) in warm.items()}
    dominant = max((k for k in per_frame if algorithmic_bytes(k, W, H, 1) > 0), key=lambda k: per_frame[k])
    # every timed kernel costs two event packets per launch: with several chunks in flight only the dominant kernel (the
    # roofline object) is timed in the timed region -- per-kernel durations are stretched by the other chunks there anyway;
    # a one-chunk run (--gops-per-gpu 1) or VP8_BENCH_TIMED=hot also times the three hot kernels of the issue-rate table
    timed_kernels = api.K_NAMES if args.profile_all else (sorted({dominant, "search1_l0", "search2", "mb"})
                                                          if os.environ.get("VP8_BENCH_TIMED", "hot" if G == 1 else "dominant") == "hot" else [dominant])
    for st in streams:
        st.enc.profile_enable(timed_kernels)
    ref_hist.update(frames=0, refs=0)

    # ---- timed region --------------------------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        streams[i % G].step()
    enqueue_s = time.perf_counter() - t0      # host time to issue everything (diagnostic: host- vs GPU-bound)
    barrier()
    elapsed = time.perf_counter() - t0
    for st in streams:
        st.enc.synchronize()   # raises if a bounded device-side wait (loop filter / intra wavefronts) expired: no number then
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    prof = {}
    for st in streams:
        for k, (ms, n) in st.enc.profile_read().items():
            pm, pn = prof.get(k, (0.0, 0))
            prof[k] = (pm + ms, pn + n)
    nrefs_avg = ref_hist["refs"] / max(ref_hist["frames"], 1)

    out = None
    if rank == 0:
        value = mbs * args.steps * world / elapsed
        ms_k, n_k = prof[dominant]
        avg_ms = ms_k / max(n_k, 1)
        abytes = algorithmic_bytes(dominant, W, H, nrefs_avg)
        achieved = abytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dominant, W, H)
        roof = {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(abytes), "launches": int(n_k),
                "note": "longest kernel per launch; none of this path's kernels is HBM-bound (integer search / transform / a "
                        "serial filter chain): the binding resource is VALU issue, reported as issue_roofline.path"}
        extra = {}
        for k, (ms, n) in prof.items():
            if k == dominant or n == 0:
                continue
            b = algorithmic_bytes(k, W, H, nrefs_avg)
            a = b / (ms / n * 1e-3) / 1e9 if b else None
            extra[k] = {"avg_launch_ms": round(ms / n, 5), "achieved_GBs": None if a is None else round(a, 3),
                        "frac": None if a is None else round(a / HBM_PEAK_GBS, 6)}
        # the resource that actually bounds the path: integer VALU issue.  Instructions per
        # launch from the committed PMC pass, durations measured live, peak from the committed micro-benchmark.
        issue = None
        insts, vt = pmc_valu(W, H, nrefs_avg)
        if insts:
            peak = vt["peak_mix_winstr_per_ns"]
            path_keys = ["search2", "search1_l0", "search1_l1", "search1_l2", "search1_l3", "search1_l4", "mb", "loop_filter",
                         "downsample", "pack", "border"] + ([] if args.host_params else ["lf_strength"])
            per_frame_insts = sum(insts[k] for k in path_keys)
            path_rate = per_frame_insts / (elapsed / (args.steps * world) * 1e9)
            issue = {"bound": "valu_issue", "unit": "wave64-instr/ns", "peak": peak, "peak_source": vt["peak_source"],
                     "path": {"instructions_per_frame": int(per_frame_insts), "achieved": round(path_rate, 1), "frac": round(path_rate / peak, 4)},
                     "instructions_source": vt["source"], "kernels": {}}
            for k in ("search2", "search1_l0", "mb"):
                if k in prof and prof[k][1]:
                    r = insts[k] / (prof[k][0] / prof[k][1] * 1e6)
                    issue["kernels"][k] = {"instructions_per_launch": int(insts[k]), "avg_launch_ms": round(prof[k][0] / prof[k][1], 5),
                                           "achieved": round(r, 1), "frac": round(r / peak, 4)}
        out = {
            "metric": "macroblocks/sec inter-frame (ME+DCT+loopfilter), 1080p", "value": round(value, 1),
            "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": f"{args.width}x{args.height} YUV420 inter frames, LAST+GOLDEN+ALTREF "
                                   f"(avg {nrefs_avg:.2f} refs/frame), loop filter on GPU, {G} GOP chunk(s) in flight per GPU",
                       "wrk_size": [W, H], "macroblocks_per_frame": mbs, "ssim_target": args.ssim_target, "qi_ladder": lastqi,
                       "altref_range": 5, "frames_per_gpu": args.steps, "gops_per_gpu": G,
                       "segment_params": "host mirror, precomputed" if args.host_params else "device, inside the step",
                       "frame_loop": "python over the C ABI" if args.host_params else "native (vp8_driver.cpp), one call per frame",
                       "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))},
            "roofline": roof,
            "issue_roofline": issue,
            "kernels_ms_per_frame_warmup": {k: round(v, 5) for k, v in sorted(per_frame.items(), key=lambda kv: -kv[1])},
            "other_kernels": extra,
            "fps": round(args.steps * world / elapsed, 2),
            "host_enqueue_ms_per_step": round(enqueue_s / args.steps * 1e3, 4),
        }
    # ---- optional: the same frames with complete VP8 frames delivered to host memory (vp8drv_get_frame: the whole
    # entropy stage on the device), one host thread per GOP chunk.  Reported next to the headline value, never as it.
    if args.bitstream and native:
        import threading
        nb = max(16, args.steps // G)
        nbytes = [0] * G

        def worker(k):
            st = streams[k]
            for _ in range(nb):
                st.drv.encode_frame_device(*st.ptrs[st.t % nd])
                st.t += 1
                nbytes[k] += len(st.drv.get_frame())

        for st in streams:   # untimed: the entropy stage allocates its scratch on first use
            st.enc.profile_enable([])   # the per-kernel HIP events of the headline leg are read already; each one is a packet more
            for _ in range(2):
                st.drv.encode_frame_device(*st.ptrs[st.t % nd])
                st.t += 1
                st.drv.get_frame()
        barrier()
        tb = time.perf_counter()
        th = [threading.Thread(target=worker, args=(k,)) for k in range(G)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        barrier()
        eb = time.perf_counter() - tb
        if dist is not None:
            tt = torch.tensor([eb], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            eb = float(tt.item())
        if rank == 0:
            out["with_bitstream"] = {"value": round(mbs * nb * G * world / eb, 1), "unit": "macroblocks/s", "fps": round(nb * G * world / eb, 1),
                                     "frames": nb * G * world, "host_threads_per_gpu": G, "avg_frame_bytes": int(sum(nbytes) / (nb * G)),
                                     "what": "native frame loop + vp8drv_get_frame: coefficient partitions and first partition coded on the "
                                             "device, finished frames in host memory (byte-identical to the reference's output)"}
    # ---- CPU baseline: the oracle on a bounded sample of the same workload (rank 0, N = 1 only) -------
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(args, host_frames, seg_last, W, H, mbs)
    if rank == 0:
        print(json.dumps(out))
    for st in streams:
        (st.drv if native else st.enc).close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(args, host_frames, seg_last, W, H, mbs):
    """Times oracle/vp8_oracle.c (the checker; OpenMP over blocks/MBs) on the host cores: kind 'port'."""
    # all host cores this process may run on (libgomp reads the variable when liboracle.so is loaded)
    os.environ["OMP_NUM_THREADS"] = str(len(os.sched_getaffinity(0)))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle_lib import Oracle
    ora = Oracle(W, H, args.ssim_target)
    threads = int(Oracle.lib().vp8o_num_threads())
    ora.upload_last(*host_frames[0])
    ora.set_segments(seg_last[1])
    # warm once with LAST only (sets golden = altref = LAST like the frame after a key frame), then time
    ora.upload_current(*host_frames[1])
    ora.inter_transform(1, 1, 0, 0)
    ora.loop_filter()
    n, t0 = 0, time.perf_counter()
    while True:
        f = host_frames[(2 + n) % len(host_frames)]
        ora.set_segments(seg_last[(2 + n) % len(host_frames)])
        ora.upload_current(*f)
        ora.inter_transform(0, 0, 1, 1)
        ora.loop_filter()
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 64:   # a bounded sample: ~12 s of host time
            break
    ora.close()
    return {"value": round(mbs * n / el, 1), "unit": "macroblocks/s", "cores": threads, "kind": "port",
            "sample": f"{n} inter frames {W}x{H}, 3 references, oracle/vp8_oracle.c with OpenMP on {threads} threads, "
                      f"{el:.1f} s"}


if __name__ == "__main__":
    main()
