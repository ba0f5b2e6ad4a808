"""The side legs that run in child processes of bench.py (a fresh process per group of legs), and their lines folded into the parent's."""
from __future__ import annotations

import json
import os
import subprocess
import sys

from .common import HBM_PEAK_GBS, algorithmic_bytes, pmc_traffic


def run_child_legs(args, dist, rdzv_key, world, script):
    """-> (the children's stdout: one JSON line per finished leg, exit status)"""
    # ---- side legs in fresh processes, BEFORE this process makes a stream: reported next to the headline value, never as it --------
    # The legs that are ONE or TWO videos coded frame after frame, and the other geometries, run in child processes: the HIP runtime
    # keeps every hardware queue a process ever used, and after 48 chunks in 8 batches a lone stream shares them badly (measured: the
    # same two-chunk leg 3 800 frames/s in a fresh process, 2 100 behind the headline's leg).  And they run FIRST, while this process
    # holds no queue: the part's scheduler keeps 24 queues resident PER DEVICE, not per process -- with the headline's eight batch
    # streams alive in the parent a child's 48-chunk leg pushed the device past that and its loop filter's waves were context-switched
    # (`waves_context_switched` 20-70 per leg in rounds 3 and 4, in a child of their own as well; 0 now).  The children of several
    # ranks form their own RCCL groups.
    child_out, child_rc = b"", 0
    run_children = not args.no_side_legs and not args.only_bitstream
    if run_children:
        if dist is not None:
            dist.barrier()
        # the children are groups of their own; they meet through files named by this run's key + the child's name (no port, no store)
        env = dict(os.environ, VP8_BENCH_CHILD="1" if (dist is not None) else "", VP8_BENCH_RDZV_KEY=rdzv_key)
        if not env["VP8_BENCH_CHILD"]:
            env.pop("VP8_BENCH_CHILD")
        # (bounded: the children rendezvous among themselves, and a child that does not come up must not hold the headline line back)
        for which in (["few"] + (["other"] if world == 1 else [])):
            argv = [a for a in sys.argv[1:] if a != "--spawn"] + ["--child-legs", which]
            try:
                child = subprocess.run([sys.executable, "-X", "faulthandler", script] + argv, env=env, stdout=subprocess.PIPE,   # (a leg that dies says where, on stderr)
                                       timeout=float(os.environ.get("VP8_BENCH_CHILD_TIMEOUT", "420")))
                child_out, child_rc = child_out + child.stdout, child_rc or child.returncode
            except subprocess.TimeoutExpired as e:
                child_out, child_rc = child_out + (e.stdout or b""), "timeout"
        if dist is not None:
            dist.barrier()
    return child_out, child_rc


def merge_child_legs(out, child_out, child_rc, dominant, W, H):
    """the children's legs into rank 0's line; the roofline fraction re-based on the dominant kernel's SOLO launch where they measured it"""
    got = 0
    for line in child_out.decode(errors="replace").splitlines():      # one line per finished leg
        try:
            part = json.loads(line)
        except Exception:
            continue
        if isinstance(part, dict):
            out.update(part)
            got += 1
    if child_rc != 0 or not got:
        out["few_stream_legs_error"] = f"child exit {child_rc} after {got} legs"
    sk = out.get("solo_kernels", {}).get("ms_per_launch", {})
    if dominant in sk:      # the roofline fraction from the kernel ALONE on the part, measured in this run (its fresh process)
        sb = algorithmic_bytes(dominant, W, H, out["solo_kernels"]["refs_per_frame"])
        sa = sb / (sk[dominant] * 1e-3) / 1e9
        roof = out["roofline"]
        roof["solo"] = {"launch_ms": sk[dominant], "algorithmic_bytes_per_launch": int(sb), "chunks_per_launch": 1, "achieved": round(sa, 3),
                        "frac": round(sa / HBM_PEAK_GBS, 6), "refs_per_frame": out["solo_kernels"]["refs_per_frame"],
                        "what": "the same kernel with the part to itself: one chunk per launch, HIP events of its own dispatch"}
        roof["achieved"], roof["frac"], roof["basis"] = roof["solo"]["achieved"], roof["solo"]["frac"], "solo launch (one chunk, the part to itself)"
        tr, _ = pmc_traffic(dominant, W, H)
        roof["traffic"] = tr       # the PMC pass ran one chunk per launch too
        ns = out["solo_kernels"]["refs_per_frame"]
        out["solo_kernels"]["hbm"] = {k: {"algorithmic_bytes": int(algorithmic_bytes(k, W, H, ns)), "achieved_GBs": round(algorithmic_bytes(k, W, H, ns) / (v * 1e-3) / 1e9, 2),
                                          "frac": round(algorithmic_bytes(k, W, H, ns) / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                                      for k, v in sk.items() if algorithmic_bytes(k, W, H, ns) > 0}
