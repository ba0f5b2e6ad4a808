"""The headline's chunks with finished VP8 frames delivered to host memory, and with the source frames coming from host memory."""
from __future__ import annotations

import os
import time

from .common import ALTREF_RANGE, CHECK_SSIM

def bitstream_leg(leg, nb, host=False):
    """the same chunks with finished VP8 frames delivered to host memory (vp8drv_get_frame: the whole entropy stage on the
    device), one host thread per GOP chunk.  host: the source frames come from page-locked host memory as well (native loop only)"""
    import threading
    G = leg.G
    nbytes = [0] * G
    checks, t_before = [None] * G, None

    def worker(k):
        d = leg.drv[k]
        for _ in range(nb):
            d.encode_frame_device(*leg.ptrs[leg.t[k] % leg.nd])
            leg.t[k] += 1
            nbytes[k] += len(d.get_frame())

    def group_worker(members, batch):     # batched launches: the group's frames in one call, then every member's bytes
        for _ in range(nb):
            leg.step_group(members, batch)
            if os.environ.get("VP8_BENCH_ENT_BATCH", "1") != "0":
                batch.get_frames_begin()
            else:
                for k in members:
                    leg.drv[k].get_frame_begin()
            for k in members:
                nbytes[k] += len(leg.drv[k].get_frame_end())

    for k in range(G):   # untimed: the entropy stage allocates its scratch on first use (sized for the densest frame: the native loop
        leg.drv[k].hip.reserve_frame_path_dense()      # starts frame t + 1 before it takes frame t's bytes, so no frame may need a second coding)
        for _ in range(2):
            leg.drv[k].encode_frame_device(*leg.ptrs[leg.t[k] % leg.nd])
            leg.t[k] += 1
            leg.drv[k].get_frame()
    leg.api.device_synchronize(leg.device)
    tb = time.perf_counter()
    if leg.batches and not os.environ.get("VP8_BENCH_PY_BITSTREAM"):
        # the native loop: a host thread per batch, every frame coded and delivered (vp8drv_batches_encode_frames_device with bytes_out)
        t_before = list(leg.t)
        _, nbo, chk = leg.api.NativeBatch.encode_frames_device_all([b for _, b in leg.batches], nb, leg.pin_host_frames() if host else leg.ptrs,
                                                                   [[leg.t[k] for k in m] for m, _ in leg.batches], frames_out="check", host=host)
        for (members, _), row, crow in zip(leg.batches, nbo, chk):
            for i, k in enumerate(members):
                leg.t[k] += nb
                nbytes[k] += row[i]
                checks[k] = crow[i]
        th = leg.batches
    else:
        th = ([threading.Thread(target=group_worker, args=(m, b)) for m, b in leg.batches] if leg.batches
              else [threading.Thread(target=worker, args=(k,)) for k in range(G)])
        for t in th:
            t.start()
        for t in th:
            t.join()
    leg.api.device_synchronize(leg.device)
    eb = time.perf_counter() - tb
    self_check = None
    if t_before is not None:
        # self-check, outside the timed region: chunk k coded again from its key frame on a driver of its own, one frame at a time; the
        # frames it delivers over the leg's span must be the leg's, byte for byte (vp8drv_frame_check folded over every frame)
        k = 5 % G
        d = leg.api.NativeDriver(leg.W, leg.H, device=leg.device, gop_size=leg.gop or (1 << 30), altref_range=ALTREF_RANGE, qi_min=0, qi_max=48,
                                 ssim_target=leg.ssim_target, device_params=1, check_ssim=CHECK_SSIM, ref_mask=3 if leg.refs == "all" else 0,
                                 conformant_stream=leg.conformant, **leg.src_kw)
        h = size = 0
        for t in range(leg.t_key[k], leg.t[k]):
            d.encode_frame_device(*leg.ptrs[t % leg.nd])
            if t >= t_before[k]:
                f = d.get_frame()
                h, size = leg.api.frame_check(h, f), size + len(f)
        d.close()
        self_check = {"chunk": k, "frames": nb, "bytes": [int(nbytes[k]), int(size)], "frame_check": [int(checks[k]), int(h)],
                      "identical": int(checks[k]) == int(h) and int(nbytes[k]) == int(size),
                      "what": "the leg's frames of one chunk against the same frames delivered by an un-batched driver of its own (every byte, in order)"}
        if not self_check["identical"] and not leg.api.load_library().vp8hip_experiments_compiled_in():     # (an experiment build leaves launches out: its line says INVALID)
            raise SystemExit(f"bench.py: self-check of the frames-out leg FAILED: {self_check}")
    return {"value": round(leg.mbs * nb * G / eb, 1), "unit": "macroblocks/s", "fps": round(nb * G / eb, 1), "frames": nb * G,
            "host_threads_per_gpu": len(th), "avg_frame_bytes": int(sum(nbytes) / (nb * G)), "self_check": self_check,
            "what": "native frame loop + vp8drv_get_frame: coefficient partitions and first partition coded on the device, finished "
                    "frames in host memory (byte-identical to the reference's output)"}



def frames_out_legs(leg, args, out, experiment_build):
    """with_bitstream and from_host_memory on the headline's own chunks (rank 0, N = 1), each followed by the chunks' reconstructions against
    the oracle's table; fills `out`"""
    # (at least 40 frames per chunk: the leg starts from an idle part with its threads 200 us apart, and over 20 frames that start
    # is 3 % of the rate -- 52.9 against 54.3 M MB/s at 40 and 54.8 at 120 on one box)
    out["with_bitstream"] = bitstream_leg(leg, max(40, args.steps))
    # ... and where the chunks stand after the leg, against the CPU oracle loop again (the frames' bytes are held against an un-batched
    # driver inside the leg; the reconstructions they leave behind against the committed table)
    oc = leg.oracle_check()
    out["with_bitstream"]["self_check_against_the_oracle"] = oc
    if oc and oc["identical"] is False and not experiment_build:
        raise SystemExit(f"bench.py: self-check FAILED -- after the frames-out leg chunks do not stand on the oracle loop's reconstruction: {oc}")
    if leg.batches and not os.environ.get("VP8_BENCH_PY_BITSTREAM"):
        # the same two loops with the host-device link in them: every source frame copied in from page-locked host memory inside the
        # timed loop (vp8hip_batch_upload_current: a copy stream per batch, two staging buffers per member), first without, then with
        # the finished frames going back; the frames' bytes are held against an un-batched driver fed from DEVICE memory, the
        # reconstructions against the oracle's table
        hs = 40      # (with the legs before it the driver's command stays inside the oracle's table of 176 frames per chunk: 166)
        leg.pin_host_frames()
        leg.run(2, host=True)        # untimed: the batches make their copy streams and staging buffers on first use
        el_h, _, _ = leg.run(hs, host=True)
        oc_in = leg.oracle_check()
        both = bitstream_leg(leg, hs, host=True)
        oc_both = leg.oracle_check()
        src_bytes = leg.source_size[0] * leg.source_size[1] * 3 // 2
        out["from_host_memory"] = {
            "frames_in": {"value": round(leg.mbs * hs * leg.G / el_h, 1), "unit": "macroblocks/s", "fps": round(hs * leg.G / el_h, 1),
                          "host_to_device_GBs": round(src_bytes * hs * leg.G / el_h / 1e9, 2), "self_check_against_the_oracle": oc_in},
            "frames_in_and_out": {"value": both["value"], "unit": "macroblocks/s", "fps": both["fps"], "avg_frame_bytes": both["avg_frame_bytes"],
                                  "host_to_device_GBs": round(src_bytes * both["fps"] / 1e9, 2), "self_check": both["self_check"],
                                  "self_check_against_the_oracle": oc_both},
            "source_bytes_per_frame": src_bytes, "frames": hs * leg.G,
            "what": "the headline's chunks and native loop with every source frame copied in from page-locked host memory inside the timed region "
                    "(vp8hip_batch_upload_current, the reference's clEnqueueWriteBuffer at vp8enc.cpp:386-388), and with the finished frames "
                    "delivered to host memory as well: the whole-job rates WITH the host-device link in them.  Never `value`."}
        for o in (oc_in, oc_both):
            if o and o["identical"] is False and not experiment_build:
                raise SystemExit(f"bench.py: self-check FAILED -- after a from_host_memory leg chunks do not stand on the oracle loop's reconstruction: {o}")
