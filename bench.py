#!/usr/bin/env python3
"""bench.py -- macroblocks/s of the inter-frame path (ME + DCT + loop filter) on N MI355X.

Workload = BASELINE.json configs[2]: 1920x1080 YUV420, LAST+GOLDEN+ALTREF, loop filter on the GPU, synthetic frames
resident in HBM, through the native frame loop behind the C ABI (vp8drv_encode_frame_device: segment data on the
device, vp8hip_inter_transform, vp8hip_loop_filter).  A rank keeps G independent closed-GOP chunks in flight
(GOPs are the unit the path shards by); one "step" = ONE INTER FRAME ON EACH OF THE G CHUNKS, so
--steps K times K*G frames per GPU.  Before anything is timed every chunk is rolled forward into GOP steady state
(>= 2*altref_range + 2 frames, chunk phases staggered), so the timed frames carry the real reference mix
(2.8 references per frame on average); the run fails if they do not.

N > 1: one process per GPU.  Under torch.distributed.run the ranks come from the environment (RANK, WORLD_SIZE, LOCAL_RANK,
MASTER_PORT; the launcher is the only thing of PyTorch involved: no rank imports it); started plainly with --gpus N > 1 (or --spawn)
this process starts the N rank processes itself, before it touches the GPU.  The ranks form the library's own process group
(vp8hip_group_*: RCCL inside libvp8hip.so, the id handed over through a file): it carries the barrier, the max-over-ranks time and
the gathering of finished frames -- there is no data-path collective, scaling is weak.  `rccl_ranks` (ncclCommCount) and
`per_rank` in the line show that N ranks met and what each measured.

Prints ONE JSON line on rank 0 with, besides the contract's fields,
  roofline       the dominant kernel ALONE on the part (solo: one chunk per launch, HIP events of its own dispatch) as algorithmic
                 bytes / time vs 8 TB/s; beside it time_shared (the launches of the timed region) and path (3.0 KB per macroblock)
  issue_roofline the resource that does bound the path: VALU issue CYCLES (per-opcode cost x PMC instruction counts)
  cpu_baseline   the CPU oracle (oracle/vp8_oracle.c, OpenMP) on a bounded sample of the same workload, and the reference's own
                 kernels compiled for x86 on one core                                                   (N = 1 only)
  with_bitstream the same frames with finished VP8 frames delivered to host memory                     (N = 1 only)
  from_host_memory  the rate WITH the host-device link in it: every source frame copied in from page-locked host memory inside the
                 timed loop (vp8hip_batch_upload_current), without and with the finished frames delivered back   (N = 1 only)
The legs that are ONE or TWO videos coded frame after frame, and the other geometries, run in a child process per rank (--child-legs;
a process keeps every hardware queue it ever used; and a leg that dies -- destroying dozens of contexts has, rarely,
ended a process inside the runtime -- takes only itself along: the child hands every finished leg over at once, the parent keeps
the headline's contexts until the line is out and leaves without tearing anything down):
  other_configs                 720p LAST-only (configs[1]), 4K 3-ref (configs[3]), 1080p SSIM target 0.93, -g 150, conformant stream (N = 1 only)
  solo_kernels, single_stream   every kernel alone / one chunk, frame after frame: MB/s and ms per frame   (N = 1 only)
  config3_literal               configs[2] as written: 300 frames, -g 150, two chunks, every frame counted (N = 1 only)
  config5_literal               configs[4] per rank: one 300-frame GOP with frames out (the next frame started before this one's
                                bytes are taken), the frames gathered on rank 0 over RCCL, the gather inside the time   (every N)
  ref_shard                     one GOP split by reference over the ranks (RCCL all_gather + broadcast per frame)     (every N)
"""
from __future__ import annotations

import argparse
import json
import os
import sys

# GPU_MAX_HW_QUEUES (the hardware queues the HIP runtime multiplexes its streams onto; default 4, streams that share a queue serialise)
# is NOT set here: libvp8hip.so sets it itself when it is loaded (16 unless the environment says otherwise; csrc/api_context.hip,
# include/vp8hip.h vp8hip_hw_queues) -- this program loads the library before anything touches the GPU, like any host that links it.

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))     # (oracle_lib: the cpu_baseline leg only)

from benchlib.common import ALTREF_RANGE, pin_to_gpu_numa_node  # noqa: E402,F401
# (the measurement scripts under scripts/ -- trace_single.py, stress_*.py ... -- reach the legs through `import bench`)
from benchlib.leg import Leg, side_leg  # noqa: E402,F401
from benchlib.legs_bitstream import bitstream_leg  # noqa: E402,F401


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120, help="timed steps; one step = one inter frame on each GOP chunk")
    ap.add_argument("--warmup", type=int, default=30, help="untimed steps after the GOP pre-roll")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--distinct-frames", type=int, default=8)
    ap.add_argument("--gops-per-gpu", type=int, default=48, help="independent GOP chunks in flight per GPU (1 = one stream)")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("VP8_BENCH_BATCH", "6")),
                    help="GOP chunks per batched launch (1 = every chunk launches its own kernels on its own stream; up to 8)")
    ap.add_argument("--refs", choices=["all", "last"], default="all", help="last = LAST only (BASELINE configs[1]: use_golden = use_altref = 0)")
    ap.add_argument("--ssim-target", type=float, default=-1.0, help="SSIM_target (reference default -1 = single LQ pass; 0.93 = the 4-pass path)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-side-legs", action="store_true", help="skip single_stream / other_configs / with_bitstream")
    ap.add_argument("--only-bitstream", action="store_true", help="of the side legs only with_bitstream (same-box A/B runs)")
    ap.add_argument("--profile-all", action="store_true", help="time every kernel with hipEvents in the timed region (adds packets)")
    ap.add_argument("--spawn", action="store_true", help="start the rank processes from here even for --gpus 1 (the N > 1 launch path)")
    ap.add_argument("--child-legs", choices=["few", "other"], default=None, help=argparse.SUPPRESS)   # internal: side legs in a fresh process
    return ap.parse_args()



def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        from benchlib.launcher import spawn_ranks
        sys.exit(spawn_ranks(args, os.path.abspath(__file__)))      # before `import torch`: the launcher never initialises the GPU
    # stdout carries ONE line, the JSON: native libraries (RCCL prints its version banner there) and anything else that
    # writes to file descriptor 1 during the run go to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VP8_BENCH_ALL_RANKS_ON_DEVICE"):     # test hook (tests/test_gpu_multirank_standin.py): several ranks on ONE GPU, over the stand-in transport
        local = int(os.environ["VP8_BENCH_ALL_RANKS_ON_DEVICE"])
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # ONE GPU runtime per process and NO GPU framework in it, at every N: the synthetic frames live in memory the library allocates
    # (vp8hip_device_alloc), and with several ranks the process group is the library's own (vp8hip_group_*, RCCL resolved by the
    # library from the ROCm it was built for).  Under torchrun only the LAUNCHER is PyTorch; the ranks never import it.
    from vp8oclenc_amd import api
    lib = api.load_library()
    if api.device_count() <= local:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dist = None
    # the rendezvous key: the same on every rank of this run, different for any other run alive on the node (the launcher's pid is the
    # ranks' common parent under torchrun; bench.py's own launcher and the parents of the child legs hand a key down)
    rdzv_key = os.environ.get("VP8_BENCH_RDZV_KEY") or f"bench-{os.getppid()}-{os.environ.get('MASTER_PORT', '0')}-{os.environ.get('TORCHELASTIC_RUN_ID', 'x')}-{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}"
    if world > 1 or os.environ.get("VP8_BENCH_CHILD") or os.environ.get("VP8_BENCH_FORCE_DIST"):
        dist = api.Group.from_env(local, rdzv_key + (f"-{args.child_legs}" if args.child_legs else ""), timeout_s=float(os.environ.get("VP8_BENCH_RDZV_TIMEOUT", "180")))
        if dist.count() != world:
            raise SystemExit(f"bench.py: RCCL counts {dist.count()} ranks in the group, WORLD_SIZE is {world}")
    # a build with the timing-experiment switches compiled in (they leave work out of launches) never prints a line
    experiment_build = bool(lib.vp8hip_experiments_compiled_in())
    if experiment_build and os.environ.get("VP8_BENCH_EXPERIMENT_BUILD", "") != "prints-an-invalid-line":
        # (scripts/ab_bitstream.sh sets the variable: the line it gets then says INVALID in its metric's name)
        raise SystemExit("bench.py: libvp8hip.so was built with -DVP8HIP_EXPERIMENTS (switches that leave work out of a launch): "
                         "rebuild it without (python -m vp8oclenc_amd.build) -- no number from this build")
    all_cpus = os.sched_getaffinity(0)
    affinity = "not pinned (VP8_BENCH_NO_PIN)" if os.environ.get("VP8_BENCH_NO_PIN", "0") not in ("", "0") else pin_to_gpu_numa_node(api, local)

    def barrier():
        if dist is not None:
            dist.barrier()
        api.device_synchronize(local)

    nd = max(2, args.distinct_frames)
    if args.child_legs:      # the fresh process of the few-stream side legs: nothing else runs here
        def emit(part):      # one line per finished leg: what is done is on its way before the next leg starts
            if rank == 0:
                os.write(json_fd, (json.dumps(part) + "\n").encode())
        from benchlib.legs_video import few_stream_legs
        few_stream_legs(args, api, dist, rank, world, local, nd, barrier, emit, which=args.child_legs)
        if dist is not None:
            dist.close()
        sys.stdout.flush()
        os.close(json_fd)
        sys.exit(0)          # (through the ordinary teardown: every context of every leg has been destroyed by now)
    # ---- side legs in fresh processes, BEFORE this process makes a stream (benchlib/children.py says why) --------------------------
    run_children = not args.no_side_legs and not args.only_bitstream
    from benchlib import children
    child_out, child_rc = children.run_child_legs(args, dist, rdzv_key, world, os.path.abspath(__file__))
    # the product north_star names -- the reference's own main() on libvp8hip.so, .y4m to .ivf, fresh processes (benchlib/drop_in.py) -- also
    # before this process holds a queue
    drop_in = None
    if rank == 0 and world == 1 and run_children and not os.environ.get("VP8_BENCH_NO_DROP_IN"):
        from benchlib.drop_in import drop_in_leg
        drop_in = drop_in_leg(args)
    # ---- the headline: warm-up, the timed region, the line's contract fields, roofline, self-checks -------------------------------
    from benchlib.headline import run_headline
    hl = run_headline(args, api, lib, dist, rank, world, local, nd, barrier, affinity, experiment_build)
    leg, out = hl.leg, hl.out
    # ---- side legs on the headline's own chunks: reported next to the headline value, never as it ---------------------------------
    if rank == 0 and world == 1 and not args.no_side_legs:
        from benchlib.legs_bitstream import frames_out_legs
        frames_out_legs(leg, args, out, experiment_build)
    host_frames = leg.host_frames
    if run_children and rank == 0:
        children.merge_child_legs(out, child_out, child_rc, hl.dominant, hl.W, hl.H)
    if drop_in is not None and rank == 0:
        out["drop_in"] = drop_in
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        os.sched_setaffinity(0, all_cpus)       # the CPU baseline gets every host core again, not the GPU's NUMA node only
        from benchlib.cpu_baseline import cpu_baseline
        out["cpu_baseline"] = cpu_baseline(args, api, host_frames, hl.W, hl.H, hl.mbs)
    leg.close()              # every context destroyed (vp8drv_batch_destroy, vp8drv_destroy -> vp8hip_destroy): the exit code is real
    if dist is not None:
        dist.close()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    sys.stderr.flush()


if __name__ == "__main__":
    main()
