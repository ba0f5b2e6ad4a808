"""One GOP chunk of the cpu_baseline leg in a process of its own: the CPU oracle (oracle/vp8_oracle.c, OpenMP on T threads, pinned to the CPUs
the parent names) codes inter frames of the bench's frames for a bounded time.  Started by benchlib/cpu_baseline.py:
    python cpu_chunk_worker.py <frames.npy> <W> <H> <nd> <phase> <threads> <seconds> <ssim_target> <cpu list>
writes "ready" when warmed up, waits for a line on stdin, codes, prints {"frames": n, "seconds": s}.  Test infrastructure (the checker timed
as a baseline; never the product)."""
import json
import os
import sys
import time


def main():
    path, W, H, nd, phase, threads, seconds, ssim_target = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), float(sys.argv[7]), float(sys.argv[8])
    cpus = [int(c) for c in sys.argv[9].split(",") if c]
    if cpus:
        os.sched_setaffinity(0, cpus)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import numpy as np
    from oracle_lib import Oracle
    from vp8oclenc_amd import api
    flat = np.load(path, mmap_mode="r")
    fsz, ysz, csz = W * H * 3 // 2, W * H, W * H // 4
    frames = [(np.ascontiguousarray(flat[i * fsz:i * fsz + ysz]).reshape(H, W), np.ascontiguousarray(flat[i * fsz + ysz:i * fsz + ysz + csz]).reshape(H // 2, W // 2),
               np.ascontiguousarray(flat[i * fsz + ysz + csz:(i + 1) * fsz]).reshape(H // 2, W // 2)) for i in range(nd)]
    lastqi, _ = api.quantizer_ladders(0, 48)
    segs = []
    for y, _, _ in frames:
        red, sharp = api.loopfilter_strength(y)
        segs.append(api.prepare_segments_data(False, lastqi, 0, red, sharp))
    ora = Oracle(W, H, ssim_target)
    Oracle.lib().vp8o_set_num_threads(threads)
    ora.upload_last(*frames[phase % nd])
    ora.set_segments(segs[(phase + 1) % nd])
    ora.upload_current(*frames[(phase + 1) % nd])
    ora.inter_transform(1, 1, 0, 0)     # golden = altref = LAST, like the frame after a key frame
    ora.loop_filter()
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    sys.stdin.readline()
    n, t0 = 0, time.perf_counter()
    while True:
        i = (phase + 2 + n) % nd
        ora.set_segments(segs[i])
        ora.upload_current(*frames[i])
        ora.inter_transform(0, 0, 1, 1)
        ora.loop_filter()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    ora.close()
    sys.stdout.write(json.dumps({"frames": n, "seconds": el}) + "\n")
    sys.stdout.flush()


if __name__ == "__main__":
    main()
