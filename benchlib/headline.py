"""The headline: G closed-GOP chunks in flight on this rank's GPU, `--steps` steps timed between barriers, the line's contract fields,
the roofline objects and the self-checks (one chunk coded again on its own; every chunk against the oracle's table)."""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

from .common import ALTREF_RANGE, CHECK_SSIM, HBM_PEAK_GBS, PREROLL, algorithmic_bytes, pmc_traffic
from .leg import Leg
from .roofline import issue_roofline


def run_headline(args, api, lib, dist, rank, world, local, nd, barrier, affinity, experiment_build):
    """returns SimpleNamespace(leg, out (rank 0's dict or None), dominant, W, H, mbs): the leg stays alive for the frames-out legs"""
    G = max(1, args.gops_per_gpu)
    B = max(1, min(8, args.batch))   # VP8HIP_MAX_BATCH
    free_before = api.device_mem_info(local)[0]
    leg = Leg(api, args.width, args.height, G, args.refs, args.ssim_target, nd, local, seed=1 + rank,
              overlap_filter=int(os.environ.get("VP8_BENCH_OVERLAP", "0")),   # experiment switch: every chunk's filter on a second stream
              batch=B)
    W, H, mbs = leg.W, leg.H, leg.mbs
    api.device_synchronize(local)
    hbm_used = free_before - api.device_mem_info(local)[0]      # contexts (surfaces, nets, coefficient buffers) + the synthetic frames

    # ---- warmup; every kernel of chunk 0 timed to find the dominant one --------------------------------------
    leg.drv[0].hip.profile_enable(api.K_NAMES)
    for d in leg.drv:
        d.hip.profile_search2_clock(True)    # k_search2 stamps its launches during the warm-up only (the stamping costs 1 %)
    leg.clock_read()
    for _ in range(max(args.warmup, 1)):
        leg.step()
    api.device_synchronize(local)
    warm = leg.drv[0].hip.profile_read()
    for d in leg.drv:
        d.hip.profile_search2_clock(False)
    leg.clock_read()
    s2_ms, s2_n = leg.s2_clock
    per_launch = {k: ms / n for k, (ms, n) in warm.items() if n}
    dominant = max((k for k in per_launch if algorithmic_bytes(k, W, H, 1) > 0), key=lambda k: per_launch[k])
    # each timed kernel costs two event packets per launch (timing four kernels on every chunk cost 6 % of the headline in
    # a same-box A/B): the timed region times only the roofline kernel, on every chunk; the other kernels' launch times
    # come from the warm-up steps of chunk 0 above (same steady state, fifteen other chunks in flight)
    timed = api.K_NAMES if args.profile_all else [dominant]
    leg.profile(timed)
    leg.clock_read()   # restart the in-kernel clock sums

    # ---- timed region: exactly --steps steps, barrier + synchronize on both sides, max over ranks -------------
    elapsed, enqueue_s, nrefs_avg = leg.run(args.steps, barrier)
    own_elapsed = elapsed
    per_rank = None
    if dist is not None:
        elapsed = dist.max(elapsed)          # the contract: the slowest rank's time
        import numpy as _np
        per_rank = dist.all_gather(_np.array([own_elapsed, float(nrefs_avg)], _np.float64))     # [world, 2] on every rank
    prof = leg.profile_read()
    clk_ms, clk_n, clk_ghz = leg.clock_read()
    for d in leg.drv:
        d.resolve()
    redone = sum(d.stats().redone_as_key for d in leg.drv)
    # ---- self-check, outside the timed region: ONE chunk of the timed run coded again from its key frame on a driver of its own
    # (no batch, no other chunk in flight); the reconstruction it ends with must be the timed chunk's.  A run whose timed frames
    # are not the frames a single un-batched encoder produces prints no line.
    verify = leg.replay_chunk((7 * (rank + 1)) % G)
    verify["against_the_oracle"] = leg.oracle_check()
    if verify["against_the_oracle"] and verify["against_the_oracle"]["identical"] is False and not experiment_build:
        raise SystemExit(f"bench.py: self-check FAILED -- chunks of the timed region do not stand on the reconstruction the CPU oracle loop reaches: {verify['against_the_oracle']}")
    if not verify["identical"] and not experiment_build:      # (an experiment build leaves work out of launches: its line says INVALID)
        raise SystemExit(f"bench.py: self-check FAILED -- the timed region's chunk {verify['chunk']} does not end where the same frames coded alone end: {verify}")
    frames_per_gpu = args.steps * G
    if args.refs == "all" and args.steps * G >= 2 * ALTREF_RANGE and nrefs_avg < 2.7:
        raise SystemExit(f"bench.py: the timed frames averaged {nrefs_avg:.2f} references per frame; LAST+GOLDEN+ALTREF in GOP "
                         "steady state is 2.8 -- this would not be BASELINE configs[2]")

    out = None
    if rank == 0:
        value = mbs * frames_per_gpu * world / elapsed
        ms_frame = elapsed / frames_per_gpu * 1e3
        ms_k, n_k = prof[dominant]
        avg_ms = ms_k / max(n_k, 1)
        items = B if (B > 1 and G > 1) else 1       # a batched launch does the stage for B chunks
        abytes = algorithmic_bytes(dominant, W, H, nrefs_avg) * items
        achieved = abytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dominant, W, H)
        traffic = None if traffic is None else traffic * items   # the PMC pass ran one chunk per launch
        path_bytes = 3000.0          # ~3.0 KB of compulsory HBM traffic per macroblock, whole inter path, 3 references
        roof = {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "basis": "time-shared launches (replaced by the solo launch where the side legs ran: see `solo`)",
                "time_shared": {"avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(abytes), "chunks_per_launch": items, "launches": int(n_k),
                                "achieved": round(achieved, 3), "frac": round(achieved / HBM_PEAK_GBS, 6),
                                "launches_in_flight": round(n_k * avg_ms * 1e-3 / elapsed, 2),
                                "what": "the dominant kernel's launches in the timed region by HIP events of their own dispatch, all chunks: with 48 "
                                        "chunks in 8 batches a launch shares the part with the other batches' kernels (launches_in_flight of this "
                                        "kernel alone), so this duration says how long a launch lasts, not how fast the kernel is"},
                "path": {"algorithmic_bytes_per_macroblock": path_bytes, "achieved": round(value * path_bytes / 1e9, 3),
                         "frac": round(value * path_bytes / 1e9 / HBM_PEAK_GBS, 6),
                         "what": "the whole inter path: SURVEY 8(d)'s compulsory bytes per macroblock x macroblocks per second of `value`"},
                "note": "None of this path's kernels is HBM-bound (integer search / transform / a serial filter chain): the binding resource is "
                        "VALU issue, see issue_roofline.  kernel_clock: the time-shared launches by the kernel's own clock"}
        lf_clock = None
        if clk_n:
            kms = clk_ms / clk_n             # every member of a batched launch stamps its own frame: this is per chunk
            cb = algorithmic_bytes("loop_filter", W, H, nrefs_avg)
            lf_clock = {"kernel": "loop_filter", "avg_ms_per_chunk": round(kms, 5), "frames": int(clk_n), "achieved": round(cb / (kms * 1e-3) / 1e9, 3),
                        "frac": round(cb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                        "how": "s_memrealtime (100 MHz) at the start of the kernel's first band and at the end of its last row",
                        "shader_clock_ghz": round(clk_ghz, 3),
                        "waves_context_switched": leg.context_switches}   # launches whose last wave changed hardware slots: 0 unless the process's queues are oversubscribed
            if dominant == "loop_filter":
                roof["kernel_clock"] = lf_clock
        if dominant == "search2" and s2_n:
            kms = s2_ms / s2_n
            roof["kernel_clock"] = {"kernel": "search2", "avg_launch_ms": round(kms, 5), "launches": int(s2_n), "achieved": round(abytes / (kms * 1e-3) / 1e9, 3),
                                    "frac": round(abytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                    "how": "s_memrealtime (100 MHz): earliest workgroup start to latest workgroup end of a launch, every 64th workgroup "
                                           "stamping (launch_clock_end, vp8hip_dev.h), over the warm-up steps (same steady state; the stamping costs "
                                           "1 % of throughput and is off in the timed region)"}
        others = {}
        for k, (ms, n) in {**warm, **prof}.items():
            if k == dominant or n == 0:
                continue
            b = algorithmic_bytes(k, W, H, nrefs_avg) * items
            a = b / (ms / n * 1e-3) / 1e9 if b else None
            others[k] = {"avg_launch_ms": round(ms / n, 5), "algorithmic_bytes_per_launch": int(b), "achieved_GBs": None if a is None else round(a, 3),
                         "frac": None if a is None else round(a / HBM_PEAK_GBS, 6)}
        out = {
            "metric": ("INVALID (experiment build: work left out of launches) " if experiment_build else "") + "macroblocks/sec inter-frame (ME+DCT+loopfilter), 1080p",
            "value": round(value, 1),
            "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": f"{args.width}x{args.height} YUV420 inter frames, {'LAST+GOLDEN+ALTREF' if args.refs == 'all' else 'LAST only'} "
                                   f"(avg {nrefs_avg:.2f} refs/frame, GOP steady state), loop filter on GPU, {G} GOP chunk(s) in flight per GPU"
                                   + (f", {B} chunks per batched launch ({(G + B - 1) // B} streams)" if B > 1 else ""),
                       "step": f"one inter frame on each of the {G} GOP chunks = {G} frames per GPU",
                       "wrk_size": [W, H], "source_size": list(leg.source_size), "padding": "on the device, inside the step (copy_with_padding)" if tuple(leg.source_size) != (W, H) else "none needed",
                       "macroblocks_per_frame": mbs, "ssim_target": args.ssim_target, "qi_ladder": list(api.quantizer_ladders(0, 48)[0]),
                       "altref_range": ALTREF_RANGE, "preroll_frames_per_chunk": f"{PREROLL}..{PREROLL + ALTREF_RANGE - 1}", "frames_per_gpu": frames_per_gpu,
                       "gops_per_gpu": G, "chunks_per_batched_launch": B, "hbm_bytes_in_use": int(hbm_used), "hbm_bytes_per_chunk": int(hbm_used // G), "refs_per_frame": round(nrefs_avg, 3),
                       "ms_per_frame": round(ms_frame, 5),
                       "segment_params": "device, inside the step", "frame_loop": "native (vp8_driver.cpp), one call per frame",
                       "check_ssim": ("on the device inside the step: intra fallback, filter update at min SSIM > 0.95, verdict read one call later"
                                      if CHECK_SSIM else "OFF (A/B run: not the reference's loop)"),
                       "frames_redone_as_key": redone, "frames_with_filter_update": None,
                       "batch_prep_stream": int(lib.vp8hip_batch_prep_mode()),     # 0 = none (default), 1 = per batch, 2 = one for all
                       "experiment_switches": "COMPILED IN" if experiment_build else "compiled out",
                       "hip_runtime_version": int(lib.vp8hip_runtime_version()), "gpu_framework_in_process": "torch" if "torch" in sys.modules else "none",
                       "process_group": None if dist is None else "vp8hip_group_* (RCCL inside libvp8hip.so; id by file rendezvous)",
                       "cpu_affinity": affinity, "hw_queues": int(lib.vp8hip_hw_queues()), "hw_queues_set_by": "the environment" if os.environ.get("GPU_MAX_HW_QUEUES") else "libvp8hip.so at load time", "launcher": "self-spawned ranks" if os.environ.get("VP8_BENCH_CHILD") else ("torchrun (launcher only)" if "TORCHELASTIC_RUN_ID" in os.environ else "single process")},
            "roofline": roof,
            "loop_filter_by_its_own_clock": lf_clock,
            "issue_roofline": issue_roofline(W, H, nrefs_avg, ms_frame, {**warm, **prof}, clk_ghz if clk_n else None),
            "kernels_ms_per_launch_warmup": {k: round(v, 5) for k, v in sorted(per_launch.items(), key=lambda kv: -kv[1])},
            "other_kernels": others,
            "fps": round(frames_per_gpu * world / elapsed, 2),
            "timed_region_s": round(elapsed, 4),
            # N ranks were seen: RCCL's own count of the communicator, and what every rank measured on its own clock
            "rccl_ranks": None if dist is None else int(dist.count()),
            "per_rank": None if per_rank is None else [{"rank": r, "timed_region_s": round(float(per_rank[r][0]), 4),
                                                        "value": round(mbs * frames_per_gpu / float(per_rank[r][0]), 1),
                                                        "refs_per_frame": round(float(per_rank[r][1]), 3)} for r in range(world)],
            "self_check": dict(verify, what="one chunk of the timed region coded again from its key frame on an un-batched driver of its own: "
                                            "CRC-32 of the final filtered reconstruction (Y, U, V), key frames; and (against_the_oracle) EVERY chunk's "
                                            "reconstruction against the committed digests of the CPU oracle loop; the run aborts on a mismatch"),
            "host_enqueue_ms_per_frame": round(enqueue_s / frames_per_gpu * 1e3, 4),
        }
    leg.profile([])
    return SimpleNamespace(leg=leg, out=out, dominant=dominant, W=W, H=H, mbs=mbs)
