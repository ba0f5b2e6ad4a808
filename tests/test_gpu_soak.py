"""The long parity runs, inside `pytest -m gpu` so that the driver's round-end run sees them (they used to live in
scripts/): BASELINE configs[2] as a 1080p GOP through the native frame loop with finished frames out, configs[3]
(4K) with all three references, one seed of the randomised end-to-end run, the launcher of bench.py and the RCCL
branch of the GOP-sharded gather.  Everything bit-exact against the CPU oracle loop + the reference's own
encode_header (oracle/_ref -> tests/bitstream_cases.expected_frame)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from oracle_lib import Oracle
from pipeline import default_segments
from vp8oclenc_amd import api
from vp8oclenc_amd.driver import InterPathDriver
from vp8oclenc_amd.synth import SynthSequence

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _all_host_cores_for_the_oracle():
    # the oracle is OpenMP code: at 1080p / 4K it needs the box's cores (the other test modules run it on 8 threads)
    lib = Oracle.lib()
    before = lib.vp8o_num_threads()
    lib.vp8o_set_num_threads(min(64, len(os.sched_getaffinity(0))))
    yield
    lib.vp8o_set_num_threads(before)


def _expected_frame():
    from bitstream_cases import expected_frame
    return expected_frame


def test_1080p_gop_three_references_native_loop_frames_out():
    """BASELINE configs[2]: 1920x1080, LAST+GOLDEN+ALTREF, loop filter on the GPU -- 30 frames of one GOP (key frame, golden,
    altref every 5) through vp8drv_*; every finished frame and every filtered reconstruction against the oracle loop."""
    expected_frame = _expected_frame()   # first partition by the reference's own encode_header where oracle/_ref travelled
    s = SynthSequence(1920, 1080, seed=11)
    W, H = s.W, s.H
    P = 8
    drv = api.NativeDriver(W, H, gop_size=150, num_partitions=P, check_ssim=1)
    ora = Oracle(W, H, -1.0)
    do = InterPathDriver(ora, W, H, gop_size=150)
    seen = set()
    for t in range(30):
        y, u, v = s.frame(t)
        drv.encode_frame_host(y, u, v)
        got = drv.get_frame()
        was_key = drv.resolve()
        out = do.encode_frame(y, u, v)
        assert was_key == (out is None), f"frame {t}: key decision differs"
        exp = expected_frame(W, H, do.last_key if out is None else out, out is None, P)
        assert got == exp, f"frame {t}: {len(got)} vs {len(exp)} bytes"
        for p_, q_ in zip(drv.hip.download_last(), ora.download_last()):
            assert np.array_equal(p_, q_), f"frame {t}: filtered reconstruction differs"
        st = drv.stats()
        if not was_key:
            seen.add((st.last_use_golden, st.last_use_altref))
    assert (1, 1) in seen and (0, 0) in seen          # frames with three references and with LAST only both occurred
    assert drv.stats().key_frames == 1
    drv.close()
    ora.close()


@pytest.mark.parametrize("W,H,frames,refs", [(1280, 720, 120, "last"), (1920, 1080, 300, "all")])
def test_baseline_configs_at_their_stated_lengths(W, H, frames, refs):
    """BASELINE configs[1] and configs[2] as long as they are written: 1280x720, 120 frames, LAST only; 1920x1080, 300 frames,
    LAST+GOLDEN+ALTREF with the reference's default -g 150 (the key frame at frame 150 included, 60 altref periods) -- through the
    native frame loop with check_SSIM and frames out, every frame's bytes and every filtered reconstruction (as a checksum) against
    the reference's loop on the CPU oracle running on all host cores."""
    import zlib
    expected_frame = _expected_frame()
    lib = Oracle.lib()
    before = lib.vp8o_num_threads()
    lib.vp8o_set_num_threads(min(128, len(os.sched_getaffinity(0))))
    s = SynthSequence(W, H, seed=17)
    mask = 3 if refs == "all" else 0
    drv = api.NativeDriver(s.W, s.H, gop_size=150, num_partitions=1, check_ssim=1, ref_mask=mask)
    ora = Oracle(s.W, s.H, -1.0)
    do = InterPathDriver(ora, s.W, s.H, gop_size=150, ref_mask=mask)
    keys, seen, updates = 0, set(), 0
    try:
        for t in range(frames):
            y, u, v = s.frame(t)
            drv.encode_frame_host(y, u, v)
            got = drv.get_frame()
            was_key = drv.resolve()
            out = do.encode_frame(y, u, v)
            assert was_key == (out is None), f"frame {t}: key decision differs"
            keys += int(was_key)
            exp = expected_frame(s.W, s.H, do.last_key if out is None else out, out is None, 1)
            assert got == exp, f"frame {t}: {len(got)} vs {len(exp)} bytes"
            a = [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in drv.hip.download_last()]
            b = [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in ora.download_last()]
            assert a == b, f"frame {t}: filtered reconstruction differs"
            if out is not None:
                seen.add((out["use_golden"], out["use_altref"]))
                updates += int(out["min_SSIM"] > np.float32(0.95))
    finally:
        lib.vp8o_set_num_threads(before)
    assert keys == (frames + 149) // 150                      # frame 0, and frame 150 of the 300
    assert seen == ({(0, 0)} if refs == "last" else {(0, 0), (1, 0), (1, 1)}), seen
    st = drv.stats()
    assert (st.key_frames, st.inter_frames) == (keys, frames - keys)
    drv.close()
    ora.close()


def test_4k_sequence_three_references_frames_out():
    """BASELINE configs[3] as a SEQUENCE: 3840x2160, 30 frames of one GOP -- key frame, golden refresh, altref periods of 5 and the
    filter's feedback from frame to frame (vp8enc.cpp:351-488) -- through the native loop with check_SSIM, frames out in 8
    partitions; every frame's bytes and every filtered reconstruction against the reference's loop on the CPU oracle."""
    import zlib
    expected_frame = _expected_frame()
    lib = Oracle.lib()
    before = lib.vp8o_num_threads()
    lib.vp8o_set_num_threads(min(128, len(os.sched_getaffinity(0))))
    s = SynthSequence(3840, 2160, seed=29)
    P = 8
    drv = api.NativeDriver(s.W, s.H, gop_size=150, num_partitions=P, check_ssim=1)
    ora = Oracle(s.W, s.H, -1.0)
    do = InterPathDriver(ora, s.W, s.H, gop_size=150)
    seen = set()
    try:
        for t in range(30):
            y, u, v = s.frame(t)
            drv.encode_frame_host(y, u, v)
            got = drv.get_frame()
            was_key = drv.resolve()
            out = do.encode_frame(y, u, v)
            assert was_key == (out is None), f"frame {t}: key decision differs"
            exp = expected_frame(s.W, s.H, do.last_key if out is None else out, out is None, P)
            assert got == exp, f"frame {t}: {len(got)} vs {len(exp)} bytes"
            a = [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in drv.hip.download_last()]
            b = [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in ora.download_last()]
            assert a == b, f"frame {t}: filtered reconstruction differs"
            if out is not None:
                seen.add((out["use_golden"], out["use_altref"]))
    finally:
        lib.vp8o_set_num_threads(before)
    assert seen == {(0, 0), (1, 0), (1, 1)}, seen       # LAST only, LAST + GOLDEN, all three
    st = drv.stats()
    assert (st.key_frames, st.inter_frames) == (1, 29)
    drv.close()
    ora.close()


def test_headline_shape_48_chunks_in_8_batches_a_thread_each_against_the_oracle():
    """What bench.py times, as it times it: 1920x1080 sources in 1920x1088 contexts, 48 closed-GOP chunks in 8 batches of 6, the
    native loop with ONE HOST THREAD PER BATCH (vp8drv_batches_encode_frames_device), check_SSIM on the device, three references,
    five inter frames per chunk with frames out.  Four of the chunks -- one in each of four different batches, at four different
    member positions -- are held against the reference's loop on the CPU oracle: the fold of every delivered frame's bytes
    (vp8drv_frame_check) and the filtered reconstruction they end with; every other chunk against the same chunk coded alone on
    an un-batched driver (the single path is what the other tests hold against the oracle)."""
    import zlib
    expected_frame = _expected_frame()
    lib = Oracle.lib()
    before = lib.vp8o_num_threads()
    lib.vp8o_set_num_threads(min(128, len(os.sched_getaffinity(0))))
    W0, H0, G, B, ND, FR = 1920, 1080, 48, 6, 8, 5
    seq = SynthSequence(W0, H0, seed=1)
    W, H = seq.W, seq.H
    src = [tuple(np.ascontiguousarray(p[:H0 // k, :W0 // k]) for p, k in zip(seq.frame(t), (1, 2, 2))) for t in range(ND)]
    padded = [tuple(np.pad(p, ((0, (H // k) - p.shape[0]), (0, 0)), mode="edge") for p, k in zip(f, (1, 2, 2))) for f in src]   # copy_with_padding
    dev = [tuple(api.to_device(p) for p in f) for f in src]
    ptrs = [tuple(p.data_ptr() for p in f) for f in dev]
    cfg = dict(gop_size=1 << 30, altref_range=2, qi_min=0, qi_max=48, device_params=1, check_ssim=1, src_width=W0, src_height=H0)
    drv = [api.NativeDriver(W, H, **cfg) for _ in range(G)]
    start = [(3 * k) % ND for k in range(G)]
    for k, d in enumerate(drv):
        assert d.encode_frame_device(*ptrs[start[k]])          # the chunk's key frame
    groups = [list(range(i, i + B)) for i in range(0, G, B)]
    batches = [api.NativeBatch([drv[k] for k in m]) for m in groups]
    try:
        _, nbytes, chk = api.NativeBatch.encode_frames_device_all(batches, FR, ptrs, [[start[k] + 1 for k in m] for m in groups], frames_out="check")
        for d in drv:
            d.resolve()
        against_oracle = [0 * B + 0, 2 * B + 3, 5 * B + 5, 7 * B + 1]
        for g, m in enumerate(groups):
            for i, k in enumerate(m):
                got_recon = [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in drv[k].hip.download_last()]
                if k in against_oracle:
                    ora = Oracle(W, H, -1.0)
                    do = InterPathDriver(ora, W, H, gop_size=1 << 30, altref_range=2)
                    h = size = 0
                    for t in range(FR + 1):
                        out = do.encode_frame(*padded[(start[k] + t) % ND])
                        assert (out is None) == (t == 0)
                        if t:
                            f = expected_frame(W, H, out, False, 1)
                            h, size = api.frame_check(h, f), size + len(f)
                    exp_recon = [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in ora.download_last()]
                    ora.close()
                else:
                    d = api.NativeDriver(W, H, **cfg)
                    h = size = 0
                    for t in range(FR + 1):
                        d.encode_frame_device(*ptrs[(start[k] + t) % ND])
                        if t:
                            f = d.get_frame()
                            h, size = api.frame_check(h, f), size + len(f)
                    d.resolve()
                    exp_recon = [zlib.crc32(np.ascontiguousarray(p).tobytes()) for p in d.hip.download_last()]
                    d.close()
                assert (nbytes[g][i], chk[g][i]) == (size, h), f"chunk {k}: the bytes of its {FR} frames differ"
                assert got_recon == exp_recon, f"chunk {k}: filtered reconstruction differs"
    finally:
        lib.vp8o_set_num_threads(before)
        for b in batches:
            b.close()
        for d in drv:
            d.close()


def test_4k_three_references_frame_pair():
    """BASELINE configs[3] geometry with LAST + GOLDEN + ALTREF: every stage tap against the oracle."""
    from test_gpu_parity import _compare, _frames, _one_frame
    f = _frames(3840, 2160, 19)
    frames = [f[2], f[0], f[1], f[3]]
    h, o = _one_frame(3840, 2160, frames, default_segments(), (1, 1), -1.0)
    _compare(h, o, [k for k in o if k in h], "4K 3 refs")
    assert len(np.unique(h["MB_reference_frame"])) >= 2


@pytest.mark.parametrize("W,H", [(7680, 4320), (8192, 16), (16, 4096), (16, 16)])
def test_beyond_the_baseline_sizes(W, H):
    """four times BASELINE's largest frame, one macroblock row / column of the largest extent, and a single macroblock: every
    stage tap of a three-reference frame against the oracle"""
    from test_gpu_parity import _compare, _frames, _one_frame
    f = _frames(W, H, 23)
    h, o = _one_frame(W, H, [f[2], f[0], f[1], f[3]], default_segments(), (1, 1), -1.0)
    _compare(h, o, [k for k in o if k in h], f"{W}x{H}")


@pytest.mark.parametrize("host_bitstream", [0, 1])
def test_8k_frames_do_not_fit_the_format_and_say_so(host_bitstream):
    """7680x4320: the first partition of a key frame (sixteen sub-block modes for each of 129 600 macroblocks) and, with this
    content, of the inter frames too (nearly every macroblock split, four vectors each) passes the 512 KB the frame tag's 19-bit
    size field can say (RFC 6386 section 9.1).  The reference writes the low 19 bits and emits a frame no decoder can read;
    here the call fails with VP8HIP_ERR_FORMAT, from the device coder and from the host one, and the encoder itself goes on:
    the reconstructions stay the oracle's."""
    s = SynthSequence(7680, 4320, seed=13)
    drv = api.NativeDriver(s.W, s.H, gop_size=150, num_partitions=8, check_ssim=1, host_bitstream=host_bitstream)
    ora = Oracle(s.W, s.H, -1.0)
    do = InterPathDriver(ora, s.W, s.H, gop_size=150)
    for t in range(2):
        y, u, v = s.frame(t)
        drv.encode_frame_host(y, u, v)
        was_key = drv.resolve()
        out = do.encode_frame(y, u, v)
        assert was_key == (out is None)
        with pytest.raises(api.Vp8HipError, match="19 bits"):
            drv.get_frame()
        for p_, q_ in zip(drv.hip.download_last(), ora.download_last()):
            assert np.array_equal(p_, q_), f"frame {t}: filtered reconstruction differs"
    drv.close()
    ora.close()


def test_fuzz_one_seed():
    """scripts/fuzz_parity.py, 12 cases of one seed: random geometry / quantizers / SSIM target / GOP / partitions / content."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_parity.py"), "--cases", "12", "--seed", "7"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "all identical" in r.stdout


def _bench(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-side-legs",
                        "--cpu-seconds", "0", *extra], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, "bench.py must print exactly ONE JSON line:\n" + r.stdout[-2000:] + r.stderr[-2000:]
    return json.loads(lines[0])


def test_bench_launcher_starts_its_own_ranks():
    """`bench.py --gpus N` without torchrun starts N rank processes itself (the parent never touches the GPU) and the
    ranks form an RCCL group; exercised here with N = 1 (--spawn) against the plain single-process run."""
    a = _bench()
    b = _bench("--spawn")
    assert a["config"]["launcher"] == "single process" and b["config"]["launcher"] == "self-spawned ranks"
    assert a["rccl_ranks"] is None and b["rccl_ranks"] == 1 and len(b["per_rank"]) == 1     # the spawned rank formed the library's RCCL group
    assert abs(b["per_rank"][0]["timed_region_s"] - b["timed_region_s"]) < 1e-3
    for d in (a, b):
        assert d["config"]["gpu_framework_in_process"] == "none" and d["config"]["hw_queues"] == 16, d["config"]
    for d in (a, b):
        g, per = d["config"]["gops_per_gpu"], d["config"]["chunks_per_batched_launch"]
        assert d["n_gpus"] == 1 and d["config"]["refs_per_frame"] >= 2.7 and d["config"]["frames_per_gpu"] == 20 * g
        assert d["roofline"]["time_shared"]["launches"] == 20 * ((g + per - 1) // per) and d["loop_filter_by_its_own_clock"]["frames"] == 20 * g
        # the process stays below the queue count at which the part's scheduler starts context-switching running waves (none when
        # the bench has the GPU to itself; a handful of the ~90 loop-filter waves per frame when it runs as a child of a test
        # process that holds queues of its own: seen once, 72 of 86 000)
        lf = d["loop_filter_by_its_own_clock"]
        assert lf["waves_context_switched"] <= 0.01 * 90 * lf["frames"] and 1.0 < lf["shader_clock_ghz"] < 3.0, lf
    # same work, same launcher-independent code path; run-to-run spread of a 20-step run on one box is about +-5 %
    assert abs(a["value"] - b["value"]) / a["value"] < 0.25, (a["value"], b["value"])


def test_bench_under_torchrun_runs_its_child_legs():
    """The driver's multi-GPU command, with one rank: `python -m torch.distributed.run ... bench.py --gpus N`.  Only the launcher is
    PyTorch: the rank and the child process of its one- and two-video legs form the library's own RCCL groups (file rendezvous keyed by
    the launcher's pid and port) and never import torch -- one HIP runtime, the one libvp8hip.so was built for, at every N."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29557", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--cpu-seconds", "0"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, VP8_BENCH_CHILD_TIMEOUT="300", VP8_BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert "few_stream_legs_error" not in d, d["few_stream_legs_error"]
    assert d["config5_literal"]["frames"] == 300 and d["ref_shard"]["value"] > 0 and d["single_stream"]["ms_per_frame"] > 0
    assert d["config"]["gpu_framework_in_process"] == "none" and d["config"]["launcher"] == "torchrun (launcher only)"
    assert d["rccl_ranks"] == 1 and len(d["per_rank"]) == 1 and d["config5_literal"]["gpu_framework_in_process"] == "none"


def test_bench_last_only_config():
    d = _bench("--refs", "last", "--width", "1280", "--height", "720")
    assert d["config"]["refs_per_frame"] == 1.0 and d["config"]["macroblocks_per_frame"] == 3600


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    from vp8oclenc_amd import api, gop_shard
    from vp8oclenc_amd.synth import SynthSequence
    grp = api.Group.from_env(0, "test-gather-%d" % os.getpid())          # RANK / WORLD_SIZE from the environment; id through a file
    assert (grp.rank, grp.world, grp.count()) == (0, 1, 1)
    seq = SynthSequence(96, 64, seed=5)
    mine = gop_shard.encode_chunks_frames(lambda: gop_shard.NativeEncoder(seq.W, seq.H, device=0), seq, gop_shard.gop_chunks(6, 3))
    plain = gop_shard.gather_frames(mine, 6)
    coll = gop_shard.gather_frames(mine, 6, grp)     # vp8hip_group_gather_bytes: ncclSend to self + ncclRecv from self, one group
    assert coll == plain and all(len(f) > 3 for f in coll)
    grp.barrier()
    assert abs(grp.max(2.5) - 2.5) < 1e-12
    rec = grp.all_gather(np.arange(5, dtype=np.float64))
    assert rec.shape == (1, 5) and np.array_equal(rec[0], np.arange(5))
    assert grp.broadcast_bytes(b"0123456789abcdef", 16) == b"0123456789abcdef"
    big = np.random.default_rng(1).integers(0, 256, 3 << 20, dtype=np.uint8)      # larger than the group's first staging buffer
    got, counts = grp.gather_bytes(big)
    assert counts.tolist() == [big.nbytes] and np.array_equal(got, big)
    grp.close()
    assert "torch" not in sys.modules
    print("rccl gather ok", sum(len(f) for f in coll))
""")


def test_gather_frames_over_rccl(tmp_path):
    """The library's own process group (vp8hip_group_*: what bench.py --gpus N and a GOP-sharded transcoder use) on one rank: file
    rendezvous, ncclCommInitRank, barrier / max / all_gather / broadcast, and gop_shard.gather_frames through vp8hip_group_gather_bytes
    -- whose code path is the same at every world size (every rank, the root included, sends; the root receives from every rank,
    itself included).  No PyTorch in the process.  The two-rank frame bookkeeping is covered on CPU in tests/test_gop_shard.py."""
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", VP8HIP_RENDEZVOUS_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl gather ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert not [f for f in os.listdir(tmp_path) if f.startswith("vp8hip-rdzv-")], "rank 0 removes the rendezvous file once the group stands"


def test_two_contexts_on_two_devices_from_worker_threads():
    """entry points select the context's device themselves (HIP's current device is per thread)."""
    if api.device_count() < 2:
        pytest.skip("needs two GPUs")
    import threading
    s = SynthSequence(128, 96, seed=3)
    drv = [api.NativeDriver(s.W, s.H, device=d, gop_size=150) for d in (0, 1)]
    out = [[], []]

    def work(i):
        for t in range(4):
            drv[i].encode_frame_host(*s.frame(t))
            out[i].append(drv[i].get_frame())

    th = [threading.Thread(target=work, args=(i,)) for i in (0, 1)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert out[0] == out[1] and len(out[0]) == 4


def test_reference_split_exchanges_inside_the_library():
    """SURVEY 8e(i) on hardware, through the C ABI: vp8hip_shard_init (a communicator of one rank: ncclCommInitRank from
    vp8hip_shard_unique_id), vp8hip_inter_search -> vp8hip_shard_share_search (RCCL broadcasts, in place in the context's nets) ->
    vp8hip_inter_finish, loop filter, vp8hip_shard_share_last -- every exchange enqueued on the context's stream with no host
    synchronisation -- and every frame compared with the plain vp8hip_inter_transform path.  The three-rank split itself:
    tests/test_ref_shard.py over gloo with the CPU oracle as the backend."""
    from vp8oclenc_amd import ref_shard
    from vp8oclenc_amd.gop_shard import frame_digest
    for (W, H, frames) in ((320, 192, 12), (1920, 1080, 4)):
        seq = SynthSequence(W, H, seed=9)
        be = ref_shard.HipRefBackend(seq.W, seq.H)
        be.shard_init(ref_shard.shard_unique_id(), 0, 1)
        assert (be.shard_rank(), be.shard_world()) == (0, 1)
        drv = ref_shard.RefShardDriver(be, None, seq.W, seq.H, altref_range=3)
        assert drv.native and drv.collective
        plain = api.Vp8Hip(seq.W, seq.H)
        pdrv = InterPathDriver(plain, seq.W, seq.H, altref_range=3, check_ssim=False)
        for t in range(frames):
            y, u, v = seq.frame(t)
            a = drv.encode_frame(y, u, v)
            b = pdrv.encode_frame(y, u, v)
            assert a["key"] == (b is None), t
            da = frame_digest(None if a["key"] else a, be.download_last())
            db = frame_digest(b, plain.download_last())
            assert da == db, (W, H, t)
        assert drv.bytes_broadcast == frames * seq.W * seq.H * 3 // 2 and drv.bytes_gathered > 0
        assert abs(be.shard_max(1.25) - 1.25) < 1e-12
        be.close(); plain.close()


def test_three_contexts_split_one_gop_by_reference_on_one_gpu_against_the_oracle():
    """What a NON-ROOT rank of the by-reference split does, executed: three HipRefBackend contexts on device 0, a thread each, rank r
    searching reference r; the vector / cost nets go owner -> vp8hip_export_search -> vp8hip_import_search on the two others (the
    landing place of vp8hip_shard_share_search's broadcasts: csrc/api_shard.hip search_nets), rank 0 finishes and filters the frame
    and its LAST goes vp8hip_export_last -> vp8hip_import_last on ranks 1 and 2 (receive_last_surface + adopt_last: the receiving
    half of vp8hip_shard_share_last).  RCCL itself refuses two ranks on one GPU, so the transport here is device-buffer hand-over
    between threads; everything behind the transport is the library's multi-rank code.  1080p, 14 frames with golden and altref
    periods: every frame's digest (modes, vectors, coefficients, filtered reconstruction) equals the plain vp8hip_inter_transform
    path AND the oracle loop -- and ranks 1 and 2, which never finish a frame, hold rank 0's LAST bit for bit at the end.
    Reference: the three per-reference queues of inter_part.h:122-135, 201-266 and the hand-over of vp8enc.cpp:395-401."""
    import threading
    from vp8oclenc_amd import ref_shard
    from vp8oclenc_amd.gop_shard import frame_digest
    FRAMES, ALTREF = 14, 3
    seq = SynthSequence(1920, 1080, seed=21)
    W, H = seq.W, seq.H
    frames = [seq.frame(t) for t in range(FRAMES)]
    grp = ref_shard.ThreadGroup(3)
    digests, lasts, errors, seen = [], [None] * 3, [], set()

    def rank_main(r):
        try:
            be = ref_shard.HipRefBackend(W, H)
            drv = ref_shard.RefShardDriver(be, grp.member(r), W, H, altref_range=ALTREF)
            assert drv.collective and not drv.native and (drv.rank, drv.world) == (r, 3)
            for t in range(FRAMES):
                o = drv.encode_frame(*frames[t])
                if r == 0:
                    digests.append(frame_digest(None if o["key"] else o, be.download_last()))
                    if not o["key"]:
                        seen.add((o["use_golden"], o["use_altref"]))
                else:
                    assert o is None
            lasts[r] = [p.copy() for p in be.download_last()]
            if r == 0:
                assert drv.bytes_broadcast == FRAMES * W * H * 3 // 2 and drv.bytes_gathered > 0
            be.close()
        except BaseException as e:      # noqa: BLE001 -- the other ranks must not wait for this one for ever
            errors.append((r, repr(e)))
            grp.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert (1, 1) in seen and (0, 0) in seen      # frames searched on all three ranks and frames with LAST only both occurred
    for r in (1, 2):
        for p_, q_ in zip(lasts[r], lasts[0]):
            assert np.array_equal(p_, q_), f"rank {r} does not hold rank 0's LAST"
    plain = api.Vp8Hip(W, H)
    pdrv = InterPathDriver(plain, W, H, altref_range=ALTREF, check_ssim=False)
    ora = Oracle(W, H, -1.0)
    odrv = InterPathDriver(ora, W, H, altref_range=ALTREF, check_ssim=False)
    for t in range(FRAMES):
        b = pdrv.encode_frame(*frames[t])
        assert frame_digest(b, plain.download_last()) == digests[t], f"frame {t}: the split differs from vp8hip_inter_transform"
        o = odrv.encode_frame(*frames[t])
        assert frame_digest(o, ora.download_last()) == digests[t], f"frame {t}: the split differs from the oracle loop"
    plain.close()
    ora.close()


def test_contexts_driven_from_worker_threads_at_once():
    """one GPU, six host threads, each with a driver of its own (different sizes, one with the intra fallback, one with frames
    through the host coder), all coding at the same time: every thread's frames are those of the same driver run alone.  The
    library keeps no state outside a context; this is what bench.py --bitstream and a transcoder with a thread per stream do."""
    import threading
    cfgs = [(320, 192, dict(num_partitions=2)), (176, 144, dict(num_partitions=1, gop_size=4)), (640, 352, dict(num_partitions=8)),
            (320, 192, dict(num_partitions=4, check_ssim=1, ssim_target=0.92, qi_min=40, qi_max=110)),
            (352, 288, dict(num_partitions=2, host_bitstream=1)), (336, 256, dict(num_partitions=1, src_width=330, src_height=250, conformant_stream=1))]
    seqs = [SynthSequence(c[2].get("src_width", c[0]) + (16 if "src_width" in c[2] else 0), c[2].get("src_height", c[1]) + (16 if "src_height" in c[2] else 0), seed=60 + i)
            for i, c in enumerate(cfgs)]
    FR = 10

    def frames_of(i):
        W, H, cfg = cfgs[i]
        sw, sh = cfg.get("src_width", W), cfg.get("src_height", H)
        return [tuple(np.ascontiguousarray(p[:sh // k, :sw // k]) for p, k in zip(seqs[i].frame(t), (1, 2, 2))) for t in range(FR)]

    inputs = [frames_of(i) for i in range(len(cfgs))]

    def run(i, out):
        W, H, cfg = cfgs[i]
        d = api.NativeDriver(W, H, **cfg)
        for f in inputs[i]:
            d.encode_frame_host(*f)
            out.append(d.get_frame())
        d.close()

    alone = [[] for _ in cfgs]
    for i in range(len(cfgs)):
        run(i, alone[i])
    for rep in range(3):
        together = [[] for _ in cfgs]
        errors = []

        def guarded(i):
            try:
                run(i, together[i])
            except Exception as e:      # noqa: BLE001 -- reported below, with the thread's number
                errors.append((i, repr(e)))

        th = [threading.Thread(target=guarded, args=(i,)) for i in range(len(cfgs))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errors, errors
        for i in range(len(cfgs)):
            assert together[i] == alone[i], (rep, i)
