"""bench.py's parts (benchlib/): what every leg shares -- constants, the ALGORITHMIC-bytes model of DESIGN.md section 4, the committed
profiler passes and oracle digests, CPU pinning."""
from __future__ import annotations

import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
# check_SSIM after every inter frame, as the reference's loop has it (vp8enc.cpp:442): the intra fallback of macroblocks below the
# SSIM target, the loop-filter update when even the worst macroblock is above 0.95, "redo as key frame".  On the device, nobody
# waiting (vp8drv_config.check_ssim with device parameters).  VP8_BENCH_CHECK=0 leaves it out (A/B runs only).
CHECK_SSIM = int(os.environ.get("VP8_BENCH_CHECK", "1"))
ALTREF_RANGE = 5
PREROLL = 2 * ALTREF_RANGE + 2


def algorithmic_bytes(kernel: str, W: int, H: int, nrefs: float) -> float:
    """ALGORITHMIC bytes per launch: mbs macroblocks, b8 = 4*mbs 8x8 blocks, nrefs references."""
    mbs = (W // 16) * (H // 16)
    b8 = 4 * mbs
    if kernel.startswith("search1_l"):
        lvl = int(kernel[-1])
        blocks = ((W >> lvl) // 8) * ((H >> lvl) // 8)
        return 133.0 * blocks * nrefs          # 64 B cur + 64 B ref + 1 B parent MV + 4 B MV out
    if kernel == "search2":
        return (64 + 64 + 4 + 4 + 4) * b8 * nrefs  # cur + ref + MV in + MV out + cost out
    if kernel == "mb":
        return (384 + 384 + 16 + 8 + 800 + 384 + 20) * mbs  # cur + ref + MVs/ref/parts in; coeffs + recon + ids out
    if kernel == "loop_filter":
        return (384 * 2 + 8) * mbs             # recon read + written in place, mask + segment id
    return 0.0


def _profile_json(name: str):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel: str, W: int, H: int):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json; FETCH_SIZE and
    WRITE_SIZE in separate passes, corrected as MI355X_MICROARCH.md prescribes).  bench.py cannot run the profiler on
    itself, so this is the last measured value for the same geometry, or None."""
    t = _profile_json("pmc_traffic.json")
    e = (t or {}).get(f"{W}x{H}", {}).get(kernel)
    return (int(e["hbm_bytes_per_launch"]), t.get("source", "profiles/pmc_traffic.json")) if e else (None, None)


def golden_digest(name: str):
    """tests/golden/full_length/<name>.json: per-frame digests of the CPU oracle loop over bench.py's own frames
    (scripts/full_length_oracle.py --oracle; committed), or None"""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "full_length", name + ".json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None



def pin_to_gpu_numa_node(api, local: int):
    """this rank's host threads onto the CPUs of its GPU's NUMA node (best effort; returns what was done, for the JSON line)"""
    try:
        bdf = api.device_pci_bus_id(local)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return f"{bdf}: no NUMA node reported"
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return f"{bdf}: node {node} has none of this process's CPUs"
        os.sched_setaffinity(0, cpus)
        return f"{bdf}: NUMA node {node}, {len(cpus)} CPUs"
    except Exception as e:      # a report, never a reason to lose the bench line
        return f"not pinned ({type(e).__name__})"

