"""N > 1 path on CPU: two gloo ranks encode alternate GOP chunks (through the same driver and the CPU
oracle backend) and together reproduce, frame for frame, what one process produces for the whole
sequence.  No data-path collective: only all_gather of the per-frame digests and a MAX all_reduce."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from oracle_lib import Oracle
from vp8oclenc_amd import gop_shard
from vp8oclenc_amd.synth import SynthSequence

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, FRAMES, GOP = 64, 48, 12, 3


def test_gop_chunks_cover_sequence_once():
    ch = gop_shard.gop_chunks(10, 4)
    assert ch == [(0, 4), (4, 4), (8, 2)]
    assert gop_shard.chunks_of_rank(10, 4, 1, 2) == [(4, 4)]
    got = sorted(sum((gop_shard.chunks_of_rank(37, 5, r, 4) for r in range(4)), []))
    assert got == gop_shard.gop_chunks(37, 5)


WORKER = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import numpy as np, torch, torch.distributed as dist
    from oracle_lib import Oracle
    from vp8oclenc_amd import gop_shard
    from vp8oclenc_amd.synth import SynthSequence
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    seq = SynthSequence({W}, {H}, seed=5)
    dist.barrier(); t0 = time.perf_counter()
    mine = gop_shard.encode_chunks(lambda: Oracle(seq.W, seq.H), seq,
                                   gop_shard.chunks_of_rank({FRAMES}, {GOP}, rank, world), seq.W, seq.H)
    dist.barrier(); el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    from torch_transport import TorchObjectGroup
    allv = gop_shard.gather_digests(mine, {FRAMES}, TorchObjectGroup(dist))
    if rank == 0:
        np.save({out!r}, allv)
    dist.destroy_process_group()
""")


def test_two_gloo_ranks_reproduce_single_process(tmp_path):
    seq = SynthSequence(W, H, seed=5)
    serial = gop_shard.encode_chunks(lambda: Oracle(seq.W, seq.H), seq, gop_shard.gop_chunks(FRAMES, GOP), seq.W, seq.H)
    serial_vec = gop_shard.gather_digests(serial, FRAMES)
    assert (serial_vec >= 0).all()
    out = str(tmp_path / "digests.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, W=W, H=H, FRAMES=FRAMES, GOP=GOP, out=out))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    sharded = np.load(out)
    assert np.array_equal(sharded, serial_vec)


class OracleEncoder:
    """CPU stand-in for gop_shard.NativeEncoder in the tests: the frame loop on the oracle, frames assembled from the
    entropy oracle's partitions and the host first-partition coder (tests/bitstream_cases.expected_frame)."""

    def __init__(self, W, H):
        from vp8oclenc_amd.driver import InterPathDriver
        self.W, self.H = W, H
        self.be = Oracle(W, H)
        self.drv = InterPathDriver(self.be, W, H, gop_size=1 << 30)

    def encode(self, y, u, v):
        from bitstream_cases import expected_frame
        out = self.drv.encode_frame(y, u, v)
        return expected_frame(self.W, self.H, self.drv.last_key if out is None else out, out is None, 2, use_reference=False)

    def close(self):
        self.be.close()


WORKER_IVF = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
    import torch.distributed as dist
    from test_gop_shard import OracleEncoder
    from vp8oclenc_amd import gop_shard
    from vp8oclenc_amd.synth import SynthSequence
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    seq = SynthSequence({W}, {H}, seed=5)
    mine = gop_shard.encode_chunks_frames(lambda: OracleEncoder(seq.W, seq.H), seq, gop_shard.chunks_of_rank({FRAMES}, {GOP}, rank, world))
    from torch_transport import TorchObjectGroup
    frames = gop_shard.gather_frames(mine, {FRAMES}, TorchObjectGroup(dist))
    if rank == 0:
        gop_shard.write_ivf({out!r}, frames, seq.W, seq.H)
    dist.destroy_process_group()
""")


def test_two_gloo_ranks_write_the_same_ivf_as_one_process(tmp_path):
    """The bitstream through the sharded path: two ranks code alternate GOP chunks, the frames are gathered in order,
    and the .ivf file is byte-identical to the one a single process writes."""
    seq = SynthSequence(W, H, seed=5)
    serial = gop_shard.encode_chunks_frames(lambda: OracleEncoder(seq.W, seq.H), seq, gop_shard.gop_chunks(FRAMES, GOP))
    one = str(tmp_path / "one.ivf")
    n = gop_shard.write_ivf(one, gop_shard.gather_frames(serial, FRAMES), seq.W, seq.H)
    data = open(one, "rb").read()
    assert len(data) == n and data[:4] == b"DKIF" and int.from_bytes(data[24:28], "little") == FRAMES + 1   # (one too many, as the reference writes it: REFERENCE_DEFECTS.md #8)
    # frame 0 of every chunk is a key frame (start code after the 3-byte tag), the others are inter frames
    off = 32
    for t in range(FRAMES):
        size = int.from_bytes(data[off:off + 4], "little")
        assert int.from_bytes(data[off + 4:off + 12], "little") == t
        body = data[off + 12:off + 12 + size]
        assert (body[0] & 1) == (0 if t % GOP == 0 else 1), t
        if t % GOP == 0:
            assert body[3:6] == bytes([0x9d, 0x01, 0x2a])
        off += 12 + size
    assert off == len(data)
    two = str(tmp_path / "two.ivf")
    script = tmp_path / "worker_ivf.py"
    script.write_text(WORKER_IVF.format(root=ROOT, W=W, H=H, FRAMES=FRAMES, GOP=GOP, out=two))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29519", str(script)],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert open(two, "rb").read() == data


def test_gather_frames_through_a_group_of_the_librarys_shape():
    """gop_shard.gather_frames_group: the frame bookkeeping around vp8hip_group_gather_bytes (api.Group on the GPU box), here with
    three thread ranks and a stand-in group that concatenates the blobs the way the library does (rank order, counts on every rank)."""
    import threading
    world, total = 3, 11
    rng = np.random.default_rng(3)
    frames = [rng.integers(0, 256, int(rng.integers(0, 400)), dtype=np.uint8).tobytes() for _ in range(total)]
    owner = [int(rng.integers(0, world)) for _ in range(total)]
    owner[4] = owner[5] = 2
    bar, slots, results = threading.Barrier(world), [None] * world, [None] * world

    class Group:
        def __init__(self, rank):
            self.rank, self.world = rank, world

        def gather_bytes(self, data, root=0):
            slots[self.rank] = bytes(data)
            bar.wait()
            counts = np.array([len(b) for b in slots], np.uint64)
            out = np.frombuffer(b"".join(slots), np.uint8).copy() if self.rank == root else None
            bar.wait()
            return out, counts

    def run(r):
        results[r] = gop_shard.gather_frames({t: frames[t] for t in range(total) if owner[t] == r}, total, Group(r), dst=1)

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert results[0] is None and results[2] is None and results[1] == frames
