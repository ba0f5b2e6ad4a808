"""The library's MULTI-RANK code executed with several ranks on the one GPU of the test box.

RCCL refuses two ranks on one device, so on a 1-GPU box the RCCL-calling branches for rank != root -- vp8hip_shard_share_search /
_share_last on a receiving rank, vp8hip_group_gather_bytes with real peers, bench.py with WORLD_SIZE = 2 -- had never run.  Here
they do: tests/standin_rccl/standin_rccl.cpp implements the eleven RCCL entry points the library resolves at run time over a
shared-memory segment (blocking, through host memory: a transport for tests, not a communication library) and is handed to the
library through its own switch VP8HIP_RCCL_LIBRARY.  Everything above those eleven calls is the product's code, unchanged.
Reference: the three per-reference queues of src/inter_part.h:122-135, 201-266 and the hand-over of src/vp8enc.cpp:395-401 (the
by-reference split); the single output file of src/encIO.h:1-30 (the gather)."""
import json
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def standin(tmp_path_factory):
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("needs g++ and the ROCm headers")
    so = str(tmp_path_factory.mktemp("standin") / "standin_rccl.so")
    subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(ROOT, "tests", "standin_rccl", "standin_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"],
                   check=True, timeout=300)
    return so


def _ranks(script_text, world, tmp_path, standin, timeout=600, extra_env=None):
    script = tmp_path / "rank.py"
    script.write_text(script_text)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", VP8HIP_RCCL_LIBRARY=standin, VP8HIP_RENDEZVOUS_DIR=str(tmp_path),
                   OMP_NUM_THREADS="8", **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}: exit {p.returncode}\n{o[-1500:]}\n{e[-3000:]}"
        assert "STAND-IN RCCL" in e, f"rank {r} did not go through the stand-in transport:\n{e[-1500:]}"
    return [o for o, _ in outs]


SPLIT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np
    from vp8oclenc_amd import api, ref_shard
    from vp8oclenc_amd.gop_shard import frame_digest
    from vp8oclenc_amd.synth import SynthSequence
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = []
    for (W, H, frames) in {cases!r}:
        seq = SynthSequence(W, H, seed=21)
        be = ref_shard.HipRefBackend(seq.W, seq.H, device=0)
        be.shard_init(api.Group.rendezvous("split-%dx%d" % (W, H), rank), rank, world)      # ncclCommInitRank (the id through a file)
        assert (be.shard_rank(), be.shard_world()) == (rank, world)
        drv = ref_shard.RefShardDriver(be, None, seq.W, seq.H, altref_range=3)
        assert drv.native and drv.collective
        for t in range(frames):
            o = drv.encode_frame(*seq.frame(t))
            if rank == 0:
                out.append(frame_digest(None if o["key"] else o, be.download_last()))
            else:
                assert o is None
        last = [p.copy() for p in be.download_last()]
        assert abs(be.shard_max(1.0 + rank) - world) < 1e-12          # ncclAllReduce(max) over the ranks
        np.save(os.path.join({tmp!r}, "last_%dx%d_rank%d.npy" % (W, H, rank)), np.concatenate([p.reshape(-1) for p in last]))
        be.close()
    if rank == 0:
        np.save(os.path.join({tmp!r}, "digests.npy"), np.array(out, np.int64))
    print("split ok")
""")


def test_by_reference_split_with_receiving_ranks_inside_the_library(tmp_path, standin):
    """vp8hip_shard_share_search / vp8hip_shard_share_last with THREE ranks (all on device 0): ranks 1 and 2 take the branches of a rank
    that is not the root -- the broadcasts land in their own nets, a free surface of their frame pool receives rank 0's planes and
    becomes their LAST -- and every frame rank 0 finishes equals the plain vp8hip_inter_transform path and (at the small size) the
    oracle loop; at the end ranks 1 and 2 hold rank 0's LAST bit for bit."""
    from oracle_lib import Oracle
    from vp8oclenc_amd import api
    from vp8oclenc_amd.driver import InterPathDriver
    from vp8oclenc_amd.gop_shard import frame_digest
    from vp8oclenc_amd.synth import SynthSequence
    cases = [(320, 192, 14), (1920, 1080, 8)]
    _ranks(SPLIT.format(root=ROOT, tmp=str(tmp_path), cases=cases), 3, tmp_path, standin)
    got = np.load(tmp_path / "digests.npy").tolist()
    at = 0
    for (W, H, frames) in cases:
        seq = SynthSequence(W, H, seed=21)
        plain = api.Vp8Hip(seq.W, seq.H)
        pdrv = InterPathDriver(plain, seq.W, seq.H, altref_range=3, check_ssim=False)
        ora = Oracle(seq.W, seq.H, -1.0) if W <= 640 else None
        odrv = InterPathDriver(ora, seq.W, seq.H, altref_range=3, check_ssim=False) if ora else None
        for t in range(frames):
            b = pdrv.encode_frame(*seq.frame(t))
            assert frame_digest(b, plain.download_last()) == got[at + t], (W, H, t, "differs from vp8hip_inter_transform")
            if odrv:
                o = odrv.encode_frame(*seq.frame(t))
                assert frame_digest(o, ora.download_last()) == got[at + t], (W, H, t, "differs from the oracle loop")
        at += frames
        lasts = [np.load(tmp_path / f"last_{W}x{H}_rank{r}.npy") for r in range(3)]
        assert np.array_equal(lasts[1], lasts[0]) and np.array_equal(lasts[2], lasts[0]), "a receiving rank does not hold rank 0's LAST"
        plain.close()
        if ora:
            ora.close()


GATHER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r})
    import numpy as np
    from vp8oclenc_amd import api, gop_shard
    from vp8oclenc_amd.synth import SynthSequence
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    grp = api.Group.from_env(0, "gather-test")
    assert (grp.rank, grp.world, grp.count()) == (rank, world, world)
    seq = SynthSequence(176, 144, seed=5)
    chunks = gop_shard.gop_chunks(12, 3)                         # four closed GOPs of three frames
    mine = gop_shard.encode_chunks_frames(lambda: gop_shard.NativeEncoder(seq.W, seq.H, device=0), seq, chunks[rank::world])
    grp.barrier()
    assert abs(grp.max(float(rank)) - (world - 1)) < 1e-12
    rec = grp.all_gather(np.array([rank, len(mine)], np.int64))
    assert rec[:, 0].tolist() == list(range(world)) and int(rec[:, 1].sum()) == 12
    assert grp.broadcast_bytes(b"from rank one!!!" if rank == 1 else None, 16, root=1) == b"from rank one!!!"
    frames = gop_shard.gather_frames(mine, 12, grp, dst=0)       # vp8hip_group_gather_bytes: every rank sends, rank 0 receives from every rank
    if rank == 0:
        serial = gop_shard.encode_chunks_frames(lambda: gop_shard.NativeEncoder(seq.W, seq.H, device=0), seq, chunks)
        assert frames == [serial[t] for t in range(12)], "the gathered sequence is not the one process's"
    else:
        assert frames is None
    grp.close()
    print("gather ok")
""")


def test_group_gathers_frames_from_real_peers(tmp_path, standin):
    """vp8hip_group_*: rendezvous by file, three ranks, barrier / max / all_gather / broadcast from a root that is not rank 0, and the
    finished frames of GOP chunks coded by three processes gathered on rank 0 in frame order -- equal to one process coding them all."""
    outs = _ranks(GATHER.format(root=ROOT), 3, tmp_path, standin)
    assert all("gather ok" in o for o in outs)


def test_bench_with_two_ranks_on_one_gpu(tmp_path, standin):
    """`bench.py --gpus 2`, both ranks on device 0 over the stand-in transport (VP8_BENCH_ALL_RANKS_ON_DEVICE: a test hook): the
    launcher, the ranks' group, the barriers around the timed region, the max over ranks, the child legs with a group of their own,
    config5_literal's gather from two ranks with each rank's frames checked against ITS committed oracle digest (ranks 0 and 1:
    tests/golden/full_length/config5_rank{0,1}.json), the by-reference split on two ranks -- everything the driver's multi-GPU run
    does except RCCL itself and the second GPU.  The rates mean nothing (two ranks share one GPU and a host-memory transport)."""
    env = dict(os.environ, VP8HIP_RCCL_LIBRARY=standin, VP8HIP_RENDEZVOUS_DIR=str(tmp_path), VP8_BENCH_ALL_RANKS_ON_DEVICE="0",
               VP8_BENCH_CHILD_TIMEOUT="600")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--cpu-seconds", "0",
                        "--gops-per-gpu", "12", "--batch", "6"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    if os.environ.get("VP8_TEST_KEEP_BENCH_LINE"):
        open(os.environ["VP8_TEST_KEEP_BENCH_LINE"], "w").write(lines[0] + "\n" + r.stderr[-6000:])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and [p["rank"] for p in d["per_rank"]] == [0, 1]
    assert d["config"]["gpu_framework_in_process"] == "none" and d["config"]["launcher"] == "self-spawned ranks"
    assert abs(d["timed_region_s"] - max(p["timed_region_s"] for p in d["per_rank"])) < 1e-3       # the slowest rank's time
    assert "few_stream_legs_error" not in d, d.get("few_stream_legs_error")
    c5 = d["config5_literal"]
    assert c5["frames"] == 600 and c5["n_gpus"] == 2 and c5["rccl_ranks"] == 2
    assert c5["self_check_against_the_oracle"]["ranks_checked"] == [0, 1] and c5["self_check_against_the_oracle"]["identical"]
    assert d["ref_shard"]["ranks"] == 2 and d["ref_shard"]["value"] > 0


def test_bench_with_eight_ranks_on_one_gpu_codes_configs4_per_rank(tmp_path, standin):
    """BASELINE configs[4] as the driver's 8-GPU run shards it -- 2400 frames of 1080p, one closed GOP of 300 frames per rank, seed 1 + rank --
    with all EIGHT ranks on device 0 over the stand-in transport: `bench.py --gpus 8` whole (launcher, the ranks' group of eight, barriers,
    max over ranks, the children's own group, the gather of 2400 finished frames on rank 0), every rank's 300 frames held against ITS
    committed oracle digest (tests/golden/full_length/config5_rank{0..7}.json).  What is left untested is eight PHYSICAL GPUs and RCCL
    itself; the rates mean nothing."""
    env = dict(os.environ, VP8HIP_RCCL_LIBRARY=standin, VP8HIP_RENDEZVOUS_DIR=str(tmp_path), VP8_BENCH_ALL_RANKS_ON_DEVICE="0",
               VP8_BENCH_CHILD_TIMEOUT="1200", VP8_BENCH_REFSHARD_FRAMES="8", VP8_BENCH_RDZV_TIMEOUT="600", VP8HIP_QUIET="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0",
                        "--gops-per-gpu", "4", "--batch", "4"], env=env, capture_output=True, text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    if os.environ.get("VP8_TEST_KEEP_BENCH_LINE"):
        open(os.environ["VP8_TEST_KEEP_BENCH_LINE"] + ".8ranks", "w").write(lines[0] + "\n" + r.stderr[-6000:])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and [p["rank"] for p in d["per_rank"]] == list(range(8))
    assert "few_stream_legs_error" not in d, d.get("few_stream_legs_error")
    c5 = d["config5_literal"]
    assert c5["frames"] == 2400 and c5["n_gpus"] == 8 and c5["rccl_ranks"] == 8 and c5["key_frames"] == 8
    oc = c5["self_check_against_the_oracle"]
    assert oc["ranks_checked"] == list(range(8)) and oc["frames_per_rank"] == 300 and oc["differing_frames"] == 0 and oc["identical"] is True
    assert d["ref_shard"]["ranks"] == 3 and d["ref_shard"]["value"] > 0
