"""Batched launches (vp8hip_batch_*, vp8drv_batch_*): four GOP chunks advanced together, one launch per stage, must produce
exactly the frames each chunk produces on its own -- the inter path and the entropy stage (vp8drv_batch_get_frame_begin)."""
import hashlib

import numpy as np
import pytest

from vp8oclenc_amd import api
from vp8oclenc_amd.synth import SynthSequence

pytestmark = pytest.mark.gpu


# extra: check_SSIM inside the batch (the reference's loop): low quantizers put the worst macroblock above 0.95 (the filter update
# on the device), an SSIM target gives replaced macroblocks and frames sent back to be key frames one call later
@pytest.mark.parametrize("W,H,n,frames,gop,extra", [(320, 192, 4, 14, 6, {}), (640, 352, 3, 8, 150, {}), (1920, 1080, 2, 5, 150, {}), (176, 144, 4, 9, 4, {}),
                                                     (320, 192, 8, 8, 5, {}), (1280, 720, 4, 6, 150, {}),
                                                     (320, 192, 4, 12, 6, dict(check_ssim=1, qi_min=0, qi_max=6)),
                                                     (320, 192, 5, 12, 150, dict(check_ssim=1, qi_min=40, qi_max=110, ssim_target=0.92)),
                                                     (640, 352, 3, 8, 5, dict(check_ssim=1, qi_min=50, qi_max=110, ssim_target=0.9)),
                                                     (1920, 1080, 3, 4, 150, dict(check_ssim=1, ssim_target=0.93))])
def test_batched_chunks_emit_the_frames_of_single_chunks(W, H, n, frames, gop, extra):
    seqs = [SynthSequence(W, H, seed=40 + i) for i in range(n)]
    Wp, Hp = seqs[0].W, seqs[0].H
    cfg = dict(gop_size=gop, altref_range=3, num_partitions=2, device_params=1, check_ssim=0)
    cfg.update(extra)
    # chunks start at different points of their GOPs (different frame types inside one batched launch): member i has
    # already coded i frames on its own when the batch takes over
    single = [api.NativeDriver(Wp, Hp, **cfg) for _ in range(n)]
    batched = [api.NativeDriver(Wp, Hp, **cfg) for _ in range(n)]
    dev = [[tuple(api.to_device(p) for p in s.frame(t)) for t in range(frames + n)] for s in seqs]
    ptr = [[tuple(p.data_ptr() for p in f) for f in d] for d in dev]
    pos = [0] * n
    for i in range(n):
        for _ in range(i):
            for drv in (single[i], batched[i]):
                drv.encode_frame_device(*ptr[i][pos[i]])
                drv.get_frame()
            pos[i] += 1
    nb = api.NativeBatch(batched)
    keys_seen = 0
    for t in range(frames):
        on = [True] * n
        if t == 2:           # one call in which the odd members sit out
            on = [i % 2 == 0 for i in range(n)]
        keys = nb.encode_frame_device([ptr[i][pos[i]] for i in range(n)], on)
        together = t % 3 != 1     # the entropy stage of the members' frames in one set of launches, or member by member
        if together:
            nb.get_frames_begin(on)
        for i in range(n):
            if not on[i]:
                assert not keys[i]
                continue
            k = single[i].encode_frame_device(*ptr[i][pos[i]])
            assert k == keys[i], (t, i)
            keys_seen += int(k)
            a, b = single[i].get_frame(), (batched[i].get_frame_end() if together else batched[i].get_frame())
            assert a == b, f"frame {t} of chunk {i}: {len(a)} vs {len(b)} bytes"
            for p_, q_ in zip(single[i].hip.download_last(), batched[i].hip.download_last()):
                assert np.array_equal(p_, q_), (t, i)
            sa, sb = single[i].stats(), batched[i].stats()
            assert (sa.last_use_golden, sa.last_use_altref, sa.inter_frames, sa.key_frames, sa.redone_as_key, sa.last_replaced) == \
                   (sb.last_use_golden, sb.last_use_altref, sb.inter_frames, sb.key_frames, sb.redone_as_key, sb.last_replaced)
            pos[i] += 1
    assert keys_seen >= (1 if gop < frames else 0)
    if extra.get("ssim_target", -1) > 0.9 and W < 1000:
        assert sum(d.stats().redone_as_key for d in batched) + sum(d.stats().last_replaced for d in batched) > 0
    nb.close()
    for d in single + batched:
        d.close()


def test_batches_on_a_host_thread_each_give_the_frames_of_single_chunks():
    """vp8drv_batches_encode_frames_device: N frames on every batch, one native host thread per batch, check_SSIM's verdicts waited for
    inside each thread -- the chunks end where the same chunks end when they are coded one by one"""
    W, H, nd, frames = 320, 192, 6, 9
    seq = SynthSequence(W, H, seed=71)
    dev = [tuple(api.to_device(p) for p in seq.frame(t)) for t in range(nd)]
    ptr = [tuple(p.data_ptr() for p in f) for f in dev]
    cfg = dict(gop_size=5, altref_range=2, num_partitions=2, device_params=1, check_ssim=1, qi_min=40, qi_max=110, ssim_target=0.92)
    starts = [[0, 2, 4], [1, 3, 5]]
    single = [[api.NativeDriver(W, H, **cfg) for _ in row] for row in starts]
    batched = [[api.NativeDriver(W, H, **cfg) for _ in row] for row in starts]
    nbs = [api.NativeBatch(row) for row in batched]
    keys, nbytes = api.NativeBatch.encode_frames_device_all(nbs, frames, ptr, starts, frames_out=True)
    for k, row in enumerate(starts):
        for i, s0 in enumerate(row):
            d = single[k][i]
            total = 0
            for t in range(frames):
                d.encode_frame_device(*ptr[(s0 + t) % nd])
                total += len(d.get_frame())
            assert total == nbytes[k][i], (k, i)        # every frame was delivered, and at the sizes of the chunk coded alone
            d.resolve()
            batched[k][i].resolve()
            for p_, q_ in zip(d.hip.download_last(), batched[k][i].hip.download_last()):
                assert np.array_equal(p_, q_), (k, i)
            a, b = d.stats(), batched[k][i].stats()
            assert (a.inter_frames, a.key_frames, a.redone_as_key, a.refs_searched) == (b.inter_frames, b.key_frames, b.redone_as_key, b.refs_searched)
            assert keys[k][i] >= 1 and a.key_frames >= keys[k][i]       # (frames sent back to be key frames are counted one call later)
    for nb in nbs:
        nb.close()
    for row in single + batched:
        for d in row:
            d.close()


def test_a_batch_members_parameter_scan_is_there_when_somebody_asks_before_the_search_launch():
    """vp8hip_batch_auto_segments leaves the scan to the frame's quarter-pel search launch; an entry point that needs the segment data
    before that launch exists (here: vp8hip_get_segments of one member, then a second frame set without any search in between) gets
    the scan launched on its own first -- the numbers of vp8hip_auto_segments on a context of its own"""
    import ctypes as C
    from vp8oclenc_amd.synth import SynthSequence
    W, H, n = 320, 192, 3
    seqs = [SynthSequence(W, H, seed=90 + i) for i in range(n)]
    lib = api.load_library()
    members = [api.Vp8Hip(seqs[0].W, seqs[0].H) for _ in range(n)]
    alone = [api.Vp8Hip(seqs[0].W, seqs[0].H) for _ in range(n)]
    hb = C.c_void_p()
    lib.vp8hip_batch_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int]
    assert lib.vp8hip_batch_create(C.byref(hb), (C.c_void_p * n)(*[m.h for m in members]), n) == 0
    lib.vp8hip_batch_set_current_device.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.vp8hip_batch_auto_segments.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int32 * 4), C.c_int]
    lib.vp8hip_batch_destroy.argtypes = [C.c_void_p]
    lib.vp8hip_batch_destroy.restype = None
    refqi = [(20 + 3 * i, 30, 40, 50 + i) for i in range(n)]
    for t in range(2):       # the second round replaces frames whose scan was asked for and (members 1, 2) never looked at
        dev = [tuple(api.to_device(p) for p in seqs[i].frame(t)) for i in range(n)]
        ptr = [[C.c_void_p(dev[i][k].data_ptr()) for i in range(n)] for k in range(3)]
        arrs = [(C.c_void_p * n)(*ptr[k]) for k in range(3)]
        assert lib.vp8hip_batch_set_current_device(hb, None, arrs[0], arrs[1], arrs[2]) == 0
        q = ((C.c_int32 * 4) * n)(*[(C.c_int32 * 4)(*r) for r in refqi])
        assert lib.vp8hip_batch_auto_segments(hb, None, (C.c_int * n)(*([0] * n)), q, 4) == 0
        for i in range(n):
            alone[i].set_current_device(*[p.data_ptr() for p in dev[i]])
            alone[i].auto_segments(False, refqi[i], 4)
        for i in ([0] if t == 0 else range(n)):
            sa, ra, ha = alone[i].get_segments()
            sb, rb, hb_ = members[i].get_segments()
            assert np.array_equal(sa, sb) and (ra, ha) == (rb, hb_), (t, i)
        api.device_synchronize()
    lib.vp8hip_batch_destroy(hb)
    for m in members + alone:
        m.close()


@pytest.mark.parametrize("W,H,src,pinned", [(320, 192, None, True), (336, 256, (330, 250), True), (640, 352, None, False)])
def test_batches_fed_from_host_memory_give_the_frames_of_batches_fed_from_device_memory(W, H, src, pinned):
    """vp8hip_batch_upload_current / vp8drv_batch_encode_frame_host / vp8drv_batches_encode_frames_host: the members' frames copied in
    from HOST memory (page-locked: asynchronous copies on the batch's copy stream, two staging buffers per member, the pack behind
    the copies and the copies behind the pack that last read their buffer; pageable: copied before the call returns) -- frame by frame
    with members sitting a call out, and through the native loop with frames out: the same bytes, the same reconstructions, as the same
    batches fed from device memory.  A source smaller than the coded size (copy_with_padding on the device, encIO.h:141-196) included.
    Reference hand-over: vp8enc.cpp:386-388."""
    nd, frames = 6, 9
    sw, sh = src if src else (W, H)
    seq = SynthSequence(sw if src else W, sh if src else H, seed=83)
    planes = []
    for t in range(nd):
        y, u, v = seq.frame(t)
        planes.append((np.ascontiguousarray(y[:sh, :sw]), np.ascontiguousarray(u[:sh // 2, :sw // 2]), np.ascontiguousarray(v[:sh // 2, :sw // 2])))
    dev = [tuple(api.to_device(p) for p in f) for f in planes]
    dptr = [tuple(p.data_ptr() for p in f) for f in dev]
    if pinned and not src:       # a frame's planes end to end in ONE page-locked buffer (one copy per frame), or a buffer per plane (three)
        host = [(api.HostBuffer(np.concatenate([p.reshape(-1) for p in f])),) for f in planes]
        hptr = [(h[0].data_ptr(), h[0].data_ptr() + f[0].size, h[0].data_ptr() + f[0].size + f[1].size) for h, f in zip(host, planes)]
    else:
        host = [tuple(api.HostBuffer(p) for p in f) for f in planes] if pinned else None
        hptr = [tuple(p.data_ptr() for p in f) for f in host] if pinned else [tuple(p.ctypes.data for p in f) for f in planes]
    cfg = dict(gop_size=5, altref_range=2, num_partitions=2, device_params=1, check_ssim=1, qi_min=30, qi_max=100, ssim_target=0.9)
    if src:
        cfg.update(src_width=sw, src_height=sh)
    starts = [[0, 2, 4], [1, 3, 5]]
    a = [[api.NativeDriver(W, H, **cfg) for _ in row] for row in starts]
    b = [[api.NativeDriver(W, H, **cfg) for _ in row] for row in starts]
    na, nb = [api.NativeBatch(r) for r in a], [api.NativeBatch(r) for r in b]
    pos = [list(r) for r in starts]
    for t in range(4):            # frame by frame; in call 2 the middle member sits out
        on = [True, t != 2, True]
        for k in range(2):
            ka = na[k].encode_frame_device([dptr[pos[k][i] % nd] for i in range(3)], on)
            kb = nb[k].encode_frame_host([hptr[pos[k][i] % nd] for i in range(3)], on)
            assert ka == kb, (t, k)
            na[k].get_frames_begin(on)
            nb[k].get_frames_begin(on)
            for i in range(3):
                if on[i]:
                    assert a[k][i].get_frame_end() == b[k][i].get_frame_end(), (t, k, i)
                    pos[k][i] += 1
    ka, ba, ca = api.NativeBatch.encode_frames_device_all(na, frames, dptr, pos, frames_out="check")
    kb, bb, cb = api.NativeBatch.encode_frames_device_all(nb, frames, hptr, pos, frames_out="check", host=True)
    assert (ka, ba, ca) == (kb, bb, cb)          # key frames, bytes and the fold over every delivered frame's bytes, per member
    kb2 = api.NativeBatch.encode_frames_device_all(nb, 3, hptr, [[p + frames for p in r] for r in pos], host=True)      # ... and without frames out
    ka2 = api.NativeBatch.encode_frames_device_all(na, 3, dptr, [[p + frames for p in r] for r in pos])
    assert ka2 == kb2
    for k in range(2):
        for i in range(3):
            a[k][i].resolve(); b[k][i].resolve()
            for p_, q_ in zip(a[k][i].hip.download_last(), b[k][i].hip.download_last()):
                assert np.array_equal(p_, q_), (k, i)
    for n_ in na + nb:
        n_.close()
    for row in a + b:
        for d in row:
            d.close()
    for f in dev + (host or []):
        for p in f:
            p.free()


@pytest.mark.parametrize("W,H,src", [(320, 192, None), (336, 256, (330, 250)), (1920, 1088, (1920, 1080))])
def test_one_video_with_the_next_frame_prefetched_from_host_memory(W, H, src):
    """vp8hip_prefetch_current / vp8drv_prefetch_frame_host: while frame t is coded, frame t + 1's planes (page-locked host memory, a frame's
    planes end to end or apart) are copied into a staging buffer of the context's; the vp8drv_encode_frame_host that names the same
    pointers packs from there.  One video with the filter on its own stream, frames out: the same bytes as the video fed from device
    memory -- with a frame that was NOT prefetched and a prefetch that is never used (other pointers follow) in between."""
    sw, sh = src if src else (W, H)
    seq = SynthSequence(sw, sh, seed=29)
    nd, frames = 5, 11
    planes = []
    for t in range(nd):
        y, u, v = seq.frame(t)
        planes.append((np.ascontiguousarray(y[:sh, :sw]), np.ascontiguousarray(u[:sh // 2, :sw // 2]), np.ascontiguousarray(v[:sh // 2, :sw // 2])))
    dev = [tuple(api.to_device(p) for p in f) for f in planes]
    dptr = [tuple(p.data_ptr() for p in f) for f in dev]
    host, hptr = [], []
    for t, f in enumerate(planes):
        if t % 2 == 0:           # end to end (one copy) ...
            h = api.HostBuffer(np.concatenate([p.reshape(-1) for p in f]))
            host.append((h,))
            hptr.append((h.data_ptr(), h.data_ptr() + f[0].size, h.data_ptr() + f[0].size + f[1].size))
        else:                    # ... or a buffer per plane (three)
            hs = tuple(api.HostBuffer(p) for p in f)
            host.append(hs)
            hptr.append(tuple(h.data_ptr() for h in hs))
    cfg = dict(gop_size=6, altref_range=2, num_partitions=2, device_params=1, check_ssim=1, overlap_filter=1)
    if src:
        cfg.update(src_width=sw, src_height=sh)
    a, b = api.NativeDriver(W, H, **cfg), api.NativeDriver(W, H, **cfg)
    for t in range(frames):
        ka = a.encode_frame_device(*dptr[t % nd])
        kb = b.encode_frame_host_ptr(*hptr[t % nd])
        if t == 3:
            b.prefetch_frame_host_ptr(*hptr[(t + 2) % nd])      # never used: frame t + 1 arrives with other pointers
        elif t != 6:                                            # (frame 7 arrives without a prefetch)
            b.prefetch_frame_host_ptr(*hptr[(t + 1) % nd])
        assert ka == kb, t
        assert a.get_frame() == b.get_frame(), t
    for p_, q_ in zip(a.hip.download_last(), b.hip.download_last()):
        assert np.array_equal(p_, q_)
    a.close(); b.close()
    for f in dev + host:
        for p in f:
            p.free()


@pytest.mark.parametrize("env", [{"VP8HIP_BATCH_S1_COARSE": "1"}, {"VP8HIP_BATCH_S1_COARSE": "2"}, {"VP8HIP_BATCH_S1_COARSE": "3"},
                                 {"VP8HIP_S1_PRE_LDS": "0"}, {"VP8HIP_S1_REF_LOOP": "0"}, {"VP8HIP_S2_SPREAD": "0"},
                                 {"VP8HIP_S2_ITER": "1"}, {"VP8HIP_S2_ITER": "2"}, {"VP8HIP_MB_PACKED": "0"}],
                         ids=lambda e: "-".join(f"{k[7:]}={v}" for k, v in e.items()))
def test_bench_with_the_switchable_forms_of_the_search_kernels_codes_the_same_frames(env):
    """The forms kept behind environment switches for same-box A/B runs (profiles/r06_*_ab.txt) -- the fused launches of a batch's hierarchical
    search (measured slower, off by default) and the round-5 forms of the two search kernels (the current ones on by default) -- are read once
    per process, so each runs in a process of its own: a small headline run whose every chunk must stand on the CPU oracle loop's table and
    whose replayed chunk must equal the batched one."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--no-side-legs", "--cpu-seconds", "0",
                        "--gops-per-gpu", "12", "--batch", "6"], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
    oc = d["self_check"]["against_the_oracle"]
    assert d["self_check"]["identical"] and oc["chunks_checked"] == 12 and oc["identical"] is True, d["self_check"]
