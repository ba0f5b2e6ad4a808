"""The legs that are ONE or TWO videos coded frame after frame, the other geometries, and the by-reference split: run in a child process
per rank (bench.py --child-legs)."""
from __future__ import annotations

import os
import sys
import time

from .common import ALTREF_RANGE, CHECK_SSIM, golden_digest
from .leg import Leg, side_leg

def literal_gops(api, W0, H0, chunks, gop_len, device, nd, refs="all", bitstream=False, seed=1, frames_out=None, frame_base=0, start=None):
    """`chunks` closed GOPs of `gop_len` frames each on this GPU, each ONE video coded frame after frame from its key frame on (loop
    filter on the chunk's second stream), one host thread per chunk, every frame counted: a BASELINE config as it is written, not
    the saturated steady state of `value`.  bitstream: every frame is also delivered as bytes (vp8drv_get_frame) into
    frames_out[frame_base + chunk * gop_len + t].  Returns (seconds, frames, key frames, frames recoded as key, bytes)."""
    import threading
    leg = Leg(api, W0, H0, 0, refs, -1.0, nd, device, seed)       # the synthetic frames in HBM; no drivers yet
    src = dict(src_width=W0, src_height=H0) if tuple(leg.source_size) != (leg.W, leg.H) else {}
    drv = [api.NativeDriver(leg.W, leg.H, device=device, gop_size=1 << 30, altref_range=ALTREF_RANGE, qi_min=0, qi_max=48, ssim_target=-1.0,
                            device_params=1, check_ssim=CHECK_SSIM, ref_mask=3 if refs == "all" else 0, overlap_filter=1, **src) for _ in range(chunks)]
    if bitstream:      # the entropy stage's scratch: not inside the timed region (the reference allocates everything in init_all)
        for d in drv:
            d.hip.reserve_frame_path_dense()
    keys, nbytes = [0] * chunks, [0] * chunks
    pipelined = bitstream and not os.environ.get("VP8_BENCH_NO_FRAME_PIPELINE")
    native_loop = pipelined and not os.environ.get("VP8_BENCH_PY_VIDEO_LOOP")
    # where the frames land: host memory allocated and touched before the clock starts (the reference's output buffers are init_all()'s)
    video_out = [d.video_out_buffer(gop_len) for d in drv] if native_loop else None

    def work(k):
        d = drv[k]
        pending = None
        if native_loop:
            # the whole loop natively (vp8drv_encode_video_device: encode(t), frame t - 1's bytes, frame t's stage, frame t's verdict):
            # the host's reaction times are on the path -- the later a frame's stage is enqueued behind its verdict, the further it
            # reaches under the next frame's LAST search
            fr, kk = d.encode_video_device(gop_len, leg.ptrs, start=3 * k, out=video_out[k], views=True)
            keys[k] += kk
            nbytes[k] += sum(len(b) for b in fr)
            if frames_out is not None:
                for t, b in enumerate(fr):
                    frames_out[frame_base + k * gop_len + t] = b
            d.hip.synchronize()
            return
        if not bitstream and not os.environ.get("VP8_BENCH_PY_VIDEO_LOOP"):
            d.encode_video_device_no_frames(gop_len, leg.ptrs, start=3 * k)     # (the same calls from C: no interpreter lock between two videos' threads)
            d.hip.synchronize()
            return
        for t in range(gop_len):
            d.encode_frame_device(*leg.ptrs[(3 * k + t) % leg.nd])
            if pipelined:
                # frame t is under way; NOW take frame t - 1's bytes (its entropy stage ran on the context's third stream beside
                # frame t - 1's loop filter and frame t's side work), then enqueue frame t's stage
                if pending is not None:
                    b = d.get_frame_end()
                    nbytes[k] += len(b)
                    if frames_out is not None:
                        frames_out[frame_base + k * gop_len + pending] = b
                d.get_frame_begin()
                pending = t
                keys[k] += int(bool(d.resolve()))
                continue
            if bitstream:
                b = d.get_frame()
                nbytes[k] += len(b)
                if frames_out is not None:
                    frames_out[frame_base + k * gop_len + t] = b
            keys[k] += int(bool(d.resolve())) if (bitstream or t == gop_len - 1) else 0
        if pending is not None:
            b = d.get_frame_end()
            nbytes[k] += len(b)
            if frames_out is not None:
                frames_out[frame_base + k * gop_len + pending] = b
        d.hip.synchronize()

    if start is not None:
        start()             # (all ranks begin their frame loops together)
    api.device_synchronize(device)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(chunks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    api.device_synchronize(device)
    el = time.perf_counter() - t0
    st = [d.stats() for d in drv]
    out = (el, chunks * gop_len, sum(s.key_frames for s in st), sum(s.redone_as_key for s in st), sum(nbytes), leg.mbs)
    for d in drv:
        d.close()
    leg.close()
    return out


def few_stream_legs(args, api, dist, rank, world, local, nd, barrier, emit=None, which="few"):
    """config5_literal (every N), ref_shard, and at N = 1 config3_literal, single_stream and other_configs: the legs that are one or two
    videos coded frame after frame, and the other geometries.  Run in a fresh process (see main()).  Returns the dict for the JSON line
    on rank 0; `emit` (if given) is also handed every finished leg at once, so that a leg that dies takes only itself along."""
    class _Out(dict):
        def __setitem__(self, k, v):
            dict.__setitem__(self, k, v)
            if emit is not None:
                emit({k: v})
    out = _Out()
    few, other = which == "few", which == "other"
    if rank == 0 and few:
        # every kernel of the path ALONE on the part: one chunk, one stream, nothing beside it, each launch timed by its own dispatch
        # (HIP events) -- the launch durations the roofline fractions are made of (with 48 chunks in flight a launch shares the part)
        solo = Leg(api, args.width, args.height, 1, args.refs, args.ssim_target, nd, local, seed=1)
        solo.profile(api.K_NAMES)
        for _ in range(4):
            solo.step()
        solo.profile_read()
        el_s, _, nrefs_s = solo.run(40)
        out["solo_kernels"] = {"refs_per_frame": round(nrefs_s, 3), "ms_per_launch": {k: round(ms / n, 5) for k, (ms, n) in solo.profile_read().items() if n},
                               "what": "one GOP chunk on one stream, every kernel timed by its own dispatch; nothing else on the part"}
        solo.close()
    if rank == 0 and world == 1 and few:
        # BASELINE configs[2] as it is written: 300 frames, the reference's -g 150 -> two closed GOPs of 150 frames, both in flight,
        # each one video coded frame after frame from its key frame on; every frame counted (2 key frames among the 300)
        # (the timed region is 0.06 s: three runs, the fastest quoted and all three shown -- one run on a box that was just given to us has read 9 % low)
        runs3 = [literal_gops(api, args.width, args.height, 2, 150, local, nd) for _ in range(3)]
        el3, n3, k3, r3, _, mbs3 = min(runs3, key=lambda r: r[0])
        out["config3_literal"] = {"workload": f"{args.width}x{args.height}, 300 frames, -g 150: two closed GOPs of 150 frames in flight on one GPU, LAST+GOLDEN+ALTREF, "
                                              "check_SSIM in the loop, loop filter on the GPU, every frame counted",
                                  "value": round(mbs3 * n3 / el3, 1), "unit": "macroblocks/s", "fps": round(n3 / el3, 1), "ms_per_frame": round(el3 / n3 * 1e3, 4),
                                  "seconds": round(el3, 4), "frames": n3, "key_frames": k3, "frames_redone_as_key": r3,
                                  "runs_fps": [round(r[1] / r[0], 1) for r in runs3], "quoted": "the fastest of three runs"}
        s1 = max(200, args.steps)
        ss = side_leg(api, args.width, args.height, 1, args.refs, args.ssim_target, s1, 20, local, nd=nd)
        ss["what"] = "ONE closed GOP coded frame after frame (what configs[2] literally is): bound by the latency of the frame's dependency chain"
        out["single_stream"] = ss
    if few:
        # BASELINE configs[4] as it is written, at every N: 300 frames per GPU = ONE closed GOP of 300 frames on each rank (2400 / 8),
        # coded end to end from its key frame with finished VP8 frames out, the frames gathered to rank 0 over RCCL in frame order
        # (gop_shard.gather_frames); the time includes the gather.  All ranks take part (collective calls).
        from vp8oclenc_amd import gop_shard
        GOP5 = int(os.environ.get("VP8_BENCH_GOP5", "300"))
        # (a 0.11 s region: every rank runs it THREE times -- the same collective calls in the same order on all ranks -- and the fastest by the
        # slowest rank's time is quoted, the usual estimator for a short region, all three shown (the later runs of a process are a few per cent
        # slower: it keeps every hardware queue it ever used); the frames checked below are the last run's)
        runs5 = []
        for _ in range(3):
            local_frames = {}
            el5, n5, k5, r5, b5, mbs5 = literal_gops(api, args.width, args.height, 1, GOP5, local, nd, bitstream=True, seed=1 + rank,
                                                     frames_out=local_frames, frame_base=rank * GOP5, start=barrier)
            # (literal_gops times its frame loop between synchronisations of its own; the clock goes on with the gather.  What is NOT in the
            # time: making the synthetic frames and creating the encoder, which is init_all() in the reference)
            t0 = time.perf_counter()
            gathered = gop_shard.gather_frames(local_frames, GOP5 * world, dist)
            barrier()
            t5 = el5 + (time.perf_counter() - t0)
            if dist is not None:
                t5 = dist.max(t5)
            runs5.append((t5, el5))
        t5, el5 = min(runs5)
        # outside the time: THIS rank's 300 frames against the CPU oracle loop's (tests/golden/full_length/config5_rank<r>.json: CRC-32
        # and length of every frame, made by scripts/full_length_oracle.py --oracle from the same synthetic frames)
        import zlib
        import numpy as np
        doc = golden_digest(f"config5_rank{rank}") if (args.width, args.height, GOP5) == (1920, 1080, 300) and CHECK_SSIM else None
        mine = [local_frames[rank * GOP5 + t] for t in range(GOP5)]
        differing = -1 if doc is None else sum((zlib.crc32(b), len(b)) != (doc["frame_crc32"][t], doc["frame_len"][t]) for t, b in enumerate(mine))
        verdicts = np.array([differing], np.int64) if dist is None else dist.all_gather(np.array([differing], np.int64)).reshape(-1)
        oracle5 = {"ranks_checked": [r for r in range(world) if verdicts[r] >= 0], "ranks_without_a_committed_digest": [r for r in range(world) if verdicts[r] < 0],
                   "frames_per_rank": GOP5, "differing_frames": int(sum(max(int(v), 0) for v in verdicts)), "identical": not any(int(v) > 0 for v in verdicts),
                   "what": "every frame a rank delivered (CRC-32 + length) against the CPU oracle loop over the same frames, tests/golden/full_length/config5_rank<r>.json"}
        if not oracle5["identical"] and not api.load_library().vp8hip_experiments_compiled_in():
            raise SystemExit(f"bench.py: config5_literal FAILED its check against the oracle digests: {oracle5}")
        if rank == 0:
            assert gathered is not None and len(gathered) == GOP5 * world and all(gathered)
            out["config5_literal"] = {
                "workload": f"{args.width}x{args.height}, {GOP5 * world} frames = one closed GOP of {GOP5} frames on each of {world} GPU(s), LAST+GOLDEN+ALTREF, check_SSIM in the loop, "
                            "finished VP8 frames gathered to rank 0 in frame order (vp8hip_group_gather_bytes: ncclSend / ncclRecv inside the library); the time includes the gather",
                "value": round(mbs5 * GOP5 * world / t5, 1), "unit": "macroblocks/s", "fps": round(GOP5 * world / t5, 1), "seconds": round(t5, 4),
                "frames": GOP5 * world, "key_frames": k5 * world, "bytes_gathered": int(sum(len(b) for b in gathered)),
                "encode_seconds_rank0": round(el5, 4), "runs_fps": [round(GOP5 * world / r[0], 1) for r in runs5], "quoted": "the fastest of three runs",
                "self_check_against_the_oracle": oracle5, "n_gpus": world, "rccl_ranks": None if dist is None else int(dist.count()),
                "gpu_framework_in_process": "torch" if "torch" in sys.modules else "none"}
        del gathered, local_frames
        # one GOP split BY REFERENCE over up to three ranks: the searches of a frame on different GPUs, vector nets
        # all_gathered, the filtered reconstruction broadcast.  Needs three ranks to mean anything; with fewer the same exchanges
        # are walked by loopback on rank 0 (what they cost on one GPU), the other ranks only keep the collectives company.
        rs = ref_shard_leg(api, dist, args.width, args.height, local, rank, world, int(os.environ.get("VP8_BENCH_REFSHARD_FRAMES", "60")))
        if rank == 0 and rs is not None:
            out["ref_shard"] = rs
    # The other geometries run in a child of their own: the one-video legs above make low-priority streams (the side stream of
    # vp8hip_filter_overlap, the entropy stage's) and RCCL brings queues too -- a process keeps every hardware queue it ever used, the
    # priority classes have queue sets of their own, and past 24 queues per process the part's scheduler rotates them and
    # context-switches running waves (`waves_context_switched` was 20-47 per leg here while these legs shared the few-stream child).
    if rank == 0 and world == 1 and other and not args.only_bitstream:
        G, B = max(1, args.gops_per_gpu), max(1, min(8, args.batch))
        oc = {}
        for name, leg_args, kw in (
                # 4K: sixteen chunks in eight batches of two (same-box: in batches of 4 55.0, of 2 59.7); 720p: batches of four
                # (48 chunks: twelve streams 102.8, eight streams 98.0)
                ("720p_last_only", (1280, 720, G, "last", -1.0, max(20, args.steps // 2), 5), dict(batch=min(B, 4))),
                ("4k_3refs", (3840, 2160, min(G, 16), "all", -1.0, max(10, args.steps // 4), 3), dict(batch=min(B, 2))),
                ("1080p_ssim93", (1920, 1080, G, "all", 0.93, max(20, args.steps // 2), 5), dict(batch=B)),
                # the reference's default GOP of 150: key frames (a raster-order wavefront each, 1.5 ms alone) among the inter frames;
                # value counts every frame
                ("1080p_gop150", (1920, 1080, G, "all", -1.0, max(20, args.steps // 2), 5), dict(batch=B, gop=150)),
                # vp8hip_conformant_stream (NOT the reference's bytes: the format's predictor, so that the stream decodes to the
                # encoder's own reconstruction): what the opt-in costs
                ("1080p_conformant_stream", (1920, 1080, G, "all", -1.0, max(20, args.steps // 2), 5), dict(batch=B, conformant=1))):
            oc[name] = side_leg(api, *leg_args, local, nd=nd, **kw)
            out["other_configs"] = dict(oc)      # (handed on after every geometry)
    return dict(out)



def ref_shard_leg(api, dist, W0, H0, local, rank, world, nframes):
    """ONE video with a frame's reference searches spread over min(world, 3) GPUs (vp8oclenc_amd/ref_shard.py; the exchanges are the
    library's: vp8hip_shard_share_search / vp8hip_shard_share_last, RCCL on the context's stream, no host synchronisation per
    frame): ms per frame.  With one rank the communicator has one member (what the calls cost on one GPU)."""
    try:
        from vp8oclenc_amd import ref_shard
        from vp8oclenc_amd.synth import SynthSequence
        members = min(world, 3)
        uid = [ref_shard.shard_unique_id() if rank == 0 else None]
        if dist is not None and world > 1:
            uid = [dist.broadcast_bytes(uid[0], ref_shard.SHARD_ID_BYTES, root=0)]     # (every rank takes part; the first three form the communicator)
        if rank >= members:
            return None
        seq = SynthSequence(W0, H0, seed=9)
        frames = [tuple(api.to_device(p, local) for p in seq.frame(t)) for t in range(6)]     # resident in HBM, like every leg's frames
        be = ref_shard.HipRefBackend(seq.W, seq.H, device=local)
        be.shard_init(uid[0], rank, members)
        drv = ref_shard.RefShardDriver(be, None, seq.W, seq.H, altref_range=ALTREF_RANGE, download=False, device_segments=True)
        for t in range(4):
            drv.encode_frame(*frames[t % len(frames)])
        be.synchronize()
        be.shard_max(0.0)          # (barrier)
        t0 = time.perf_counter()
        for t in range(4, 4 + nframes):
            drv.encode_frame(*frames[t % len(frames)])
        be.synchronize()
        el = be.shard_max(time.perf_counter() - t0)
        mbs = (seq.W // 16) * (seq.H // 16)
        out = {"workload": f"{W0}x{H0}, one GOP, a frame's LAST / GOLDEN / ALTREF searches on " + (f"{members} GPUs" if members > 1 else "one GPU (a communicator of one rank)"),
               "value": round(mbs * nframes / el, 1), "unit": "macroblocks/s", "ms_per_frame": round(el / nframes * 1e3, 4), "frames": nframes,
               "bytes_of_nets_shared_per_frame": int(drv.bytes_gathered / (nframes + 4)), "bytes_broadcast_per_frame": int(drv.bytes_broadcast / (nframes + 4)),
               "ranks": members,
               "what": "vp8hip_inter_search on every rank's references, vp8hip_shard_share_search (one group of RCCL broadcasts, in place in the "
                       "context's nets), vp8hip_inter_finish + loop filter on rank 0, vp8hip_shard_share_last (the padded planes out of rank 0's "
                       "frame pool into the others'): all on the context's stream, no host synchronisation per frame; frames resident in HBM, "
                       "segment data on the device"}
        be.close()
        return out
    except Exception as e:      # a side leg is a report, never a reason to lose the bench line
        return {"error": repr(e)[:300]}

