"""one GOP split by reference over three ranks (vp8oclenc_amd.ref_shard) -- three gloo ranks on the CPU
oracle, each searching one of LAST / GOLDEN / ALTREF, all_gather of vectors + costs, rank 0 finishing the frame and
broadcasting the reconstruction, reproduce frame for frame what one process codes."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from oracle_lib import Oracle
from refshard_cpu import OracleRefBackend
from vp8oclenc_amd import ref_shard
from vp8oclenc_amd.driver import InterPathDriver
from vp8oclenc_amd.gop_shard import frame_digest
from vp8oclenc_amd.synth import SynthSequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, FRAMES = 64, 48, 9


def serial_digests():
    seq = SynthSequence(W, H, seed=7)
    ora = Oracle(seq.W, seq.H)
    drv = InterPathDriver(ora, seq.W, seq.H, gop_size=150, altref_range=3, check_ssim=False)
    out = []
    for t in range(FRAMES):
        o = drv.encode_frame(*seq.frame(t))
        out.append(frame_digest(o, ora.download_last()))
    ora.close()
    return np.array(out, np.int64)


def test_one_process_backend_matches_the_frame_driver():
    """the CPU backend + RefShardDriver without any collective == InterPathDriver on the oracle's whole-frame driver"""
    seq = SynthSequence(W, H, seed=7)
    be = OracleRefBackend(seq.W, seq.H)
    drv = ref_shard.RefShardDriver(be, None, seq.W, seq.H, altref_range=3)
    got = []
    refs_used = set()
    for t in range(FRAMES):
        o = drv.encode_frame(*seq.frame(t))
        if not o["key"]:
            refs_used.add((o["use_golden"], o["use_altref"]))
        got.append(frame_digest(None if o["key"] else o, be.download_last()))
    assert np.array_equal(np.array(got, np.int64), serial_digests())
    assert (1, 1) in refs_used


WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch.distributed as dist
    from refshard_cpu import OracleRefBackend
    from vp8oclenc_amd import ref_shard
    from vp8oclenc_amd.gop_shard import frame_digest
    from vp8oclenc_amd.synth import SynthSequence
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    seq = SynthSequence({W}, {H}, seed=7)
    be = OracleRefBackend(seq.W, seq.H)
    from torch_transport import TorchObjectGroup
    drv = ref_shard.RefShardDriver(be, TorchObjectGroup(dist), seq.W, seq.H, altref_range=3)
    got = []
    for t in range({FRAMES}):
        o = drv.encode_frame(*seq.frame(t))
        if dist.get_rank() == 0:
            got.append(frame_digest(None if o["key"] else o, be.download_last()))
        else:
            assert o is None
    if dist.get_rank() == 0:
        assert drv.bytes_gathered > 0 and drv.bytes_broadcast == {FRAMES} * seq.W * seq.H * 3 // 2
        np.save({out!r}, np.array(got, np.int64))
    dist.destroy_process_group()
""")


def test_three_gloo_ranks_split_by_reference_reproduce_one_process(tmp_path):
    out = str(tmp_path / "digests.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, W=W, H=H, FRAMES=FRAMES, out=out))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
                        "--master-port", "29523", str(script)], env=dict(os.environ, OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert np.array_equal(np.load(out), serial_digests())


def test_three_thread_ranks_exchange_objects_and_reproduce_one_process():
    """ref_shard.ThreadGroup: the ranks as threads of one process, the exchanges as objects changing hands (on the GPU box: device
    buffers through vp8hip_export_* / vp8hip_import_*, tests/test_gpu_soak.py) -- here with the CPU oracle as the backend."""
    import threading
    grp = ref_shard.ThreadGroup(3)
    got, errors = [], []

    def rank_main(r):
        try:
            seq = SynthSequence(W, H, seed=7)
            be = OracleRefBackend(seq.W, seq.H)
            drv = ref_shard.RefShardDriver(be, grp.member(r), seq.W, seq.H, altref_range=3)
            assert drv.collective and (drv.rank, drv.world) == (r, 3)
            for t in range(FRAMES):
                o = drv.encode_frame(*seq.frame(t))
                if r == 0:
                    got.append(frame_digest(None if o["key"] else o, be.download_last()))
                else:
                    assert o is None
        except BaseException as e:      # noqa: BLE001
            errors.append((r, repr(e)))
            grp.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(3)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert np.array_equal(np.array(got, np.int64), serial_digests())
