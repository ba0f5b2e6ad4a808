#!/usr/bin/env python3
"""Encode a sequence to .ivf with the native frame loop, GOP chunks sharded over the visible GPUs.

    python scripts/encode_ivf.py out.ivf [--y4m in.y4m | --yuv in.yuv --width 1920 --height 1080] [--frames 120 --gop 30 --partitions 4]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/encode_ivf.py out.ivf ...

--y4m: YUV4MPEG2, the reference's own input format (size and frame rate from its header, vp8oclenc_amd/y4m.py); --yuv: raw
I420 of the given size; without either the synthetic sequence of the tests is used.  A raw I420 file has the given width/height (even numbers); when
they are not multiples of 16 the frames are padded on the device (vp8hip_set_source_size = copy_with_padding, encIO.h:141-196)
and the key frames carry the source size as display size."""
import argparse, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vp8oclenc_amd import gop_shard
from vp8oclenc_amd.synth import SynthSequence


class YuvFile:
    def __init__(self, path, W, H):
        self.W, self.H = W, H
        self.m = np.memmap(path, np.uint8, "r")
        self.fsz = W * H * 3 // 2
        self.n = len(self.m) // self.fsz

    def frame(self, t):
        b = self.m[t * self.fsz:(t + 1) * self.fsz]
        W, H = self.W, self.H
        return (np.ascontiguousarray(b[:W * H].reshape(H, W)), np.ascontiguousarray(b[W * H:W * H * 5 // 4].reshape(H // 2, W // 2)),
                np.ascontiguousarray(b[W * H * 5 // 4:].reshape(H // 2, W // 2)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out"); ap.add_argument("--yuv"); ap.add_argument("--y4m")
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=60); ap.add_argument("--gop", type=int, default=30)
    ap.add_argument("--partitions", type=int, default=1); ap.add_argument("--qmin", type=int, default=0); ap.add_argument("--qmax", type=int, default=48)
    ap.add_argument("--ssim-target", type=float, default=-1.0); ap.add_argument("--framerate", type=int, default=30)
    ap.add_argument("--conformant", action="store_true", help="vp8hip_conformant_stream: NOT the reference byte for byte, but a stream that decodes to the encoder's own reconstruction")
    a = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:      # the library's own process group (vp8hip_group_*: RCCL inside libvp8hip.so, the id through a file): no GPU framework in the host
        from vp8oclenc_amd import api
        dist = api.Group.from_env(local, f"encode_ivf-{os.getppid()}-{os.environ.get('MASTER_PORT', '0')}")
    if a.y4m:
        from vp8oclenc_amd.y4m import Y4mFile
        seq = Y4mFile(a.y4m)
        a.framerate = seq.framerate or a.framerate
    else:
        seq = YuvFile(a.yuv, a.width, a.height) if a.yuv else SynthSequence(a.width, a.height, seed=1)
    frames = min(a.frames, seq.n) if (a.yuv or a.y4m) else a.frames
    Wc, Hc = (seq.W + 15) // 16 * 16, (seq.H + 15) // 16 * 16            # the coded ("wrk") size, init.h:375-392
    src = dict(src_width=seq.W, src_height=seq.H) if (Wc, Hc) != (seq.W, seq.H) else {}
    t0 = time.perf_counter()
    mine = gop_shard.encode_chunks_frames(
        lambda: gop_shard.NativeEncoder(Wc, Hc, device=local, **src, num_partitions=a.partitions, qi_min=a.qmin, qi_max=a.qmax,
                                        ssim_target=a.ssim_target, check_ssim=1, conformant_stream=int(a.conformant)),
        seq, gop_shard.chunks_of_rank(frames, a.gop, rank, world))
    allf = gop_shard.gather_frames(mine, frames, dist)
    if rank == 0:
        n = gop_shard.write_ivf(a.out, allf, seq.W, seq.H, a.framerate)
        el = time.perf_counter() - t0
        print(f"{a.out}: {frames} frames {seq.W}x{seq.H}, {n} bytes, {world} GPU(s), {frames / el:.1f} frames/s including the host-side frame source")
    if dist is not None:
        dist.close()


if __name__ == "__main__":
    main()
