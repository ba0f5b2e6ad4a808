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
import socket
import subprocess
import sys
import time

# GPU_MAX_HW_QUEUES (the hardware queues the HIP runtime multiplexes its streams onto; default 4, streams that share a queue serialise)
# is NOT set here: libvp8hip.so sets it itself when it is loaded (16 unless the environment says otherwise; csrc/api_context.hip,
# include/vp8hip.h vp8hip_hw_queues) -- this program loads the library before anything touches the GPU, like any host that links it.

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

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


# ---------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no torchrun around it starts its own ranks.  Nothing here touches HIP.
def spawn_ranks(args) -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    argv = [a for a in sys.argv[1:] if a != "--spawn"]
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VP8_BENCH_CHILD="1", VP8_BENCH_RDZV_KEY=f"bench-{os.getpid()}-{port}", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def golden_digest(name: str):
    """tests/golden/full_length/<name>.json: per-frame digests of the CPU oracle loop over bench.py's own frames
    (scripts/full_length_oracle.py --oracle; committed), or None"""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "full_length", name + ".json")) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


# ---------------------------------------------------------------------------------------------------------------
class Leg:
    """G GOP chunks of one geometry on one GPU: frames in HBM, native drivers, pre-rolled to GOP steady state."""

    def __init__(self, api, W0, H0, G, refs, ssim_target, nd, device, seed, overlap_filter=0, batch=1, gop=None, conformant=0):
        from vp8oclenc_amd.synth import bench_frames
        self.api, self.device = api, device
        self.overlap_filter, self.ssim_target, self.gop, self.conformant, self.seed0 = overlap_filter, ssim_target, gop, conformant, seed
        # a source below the coded size (1920x1080 in a 1920x1088 context) is handed over as it is: copy_with_padding
        # (encIO.h:141-196) runs inside the launch that takes a frame in, i.e. inside every timed step (vp8hip_set_source_size);
        # host_frames = the same frames padded on the host: what the CPU baseline codes
        self.W, self.H, source, self.host_frames = bench_frames(W0, H0, seed, nd)
        self.mbs = (self.W // 16) * (self.H // 16)
        self.G, self.nd, self.refs = G, nd, refs
        src = dict(src_width=W0, src_height=H0) if source is not self.host_frames else {}
        self.source_size = (W0, H0) if src else (self.W, self.H)
        self.src_kw = src
        self.dev_frames = [tuple(api.to_device(p, device) for p in f) for f in source]     # (the library's own allocator: no second GPU runtime in the process)
        self.ptrs = [tuple(p.data_ptr() for p in f) for f in self.dev_frames]
        self.source, self.pinned, self.host_ptrs = source, [], None
        self.drv, self.t, self.batches = [], [], []
        self.frames = self.refsum = self.keys = 0
        if G == 0:
            return          # (the synthetic frames only: literal_gops brings its own drivers)
        for k in range(G):
            d = api.NativeDriver(self.W, self.H, device=device, gop_size=gop or (1 << 30), altref_range=ALTREF_RANGE, qi_min=0, qi_max=48,
                                 ssim_target=ssim_target, device_params=1, check_ssim=CHECK_SSIM, ref_mask=3 if refs == "all" else 0,
                                 overlap_filter=overlap_filter, conformant_stream=conformant, **src)
            t = (k * 3) % nd                                   # chunks start at different frames of the sequence
            assert d.encode_frame_device(*self.ptrs[t % nd])   # frame 0 of the chunk: key frame
            self.drv.append(d)
            self.t.append(t + 1)
            self.t_key = getattr(self, "t_key", []) + [t]
            # batched launches: groups of `batch` chunks advance together, one launch per stage for the group (vp8drv_batch_*).
            # A group is formed as soon as its members exist: their own streams go and the group gets a new one, so the
            # process never holds more than batch + G / batch streams and every group ends up on a hardware queue of its own
            # (the runtime hands a new stream the least used of its queues)
            if batch > 1 and (len(self.drv) % batch == 0 or k == G - 1):
                k0 = len(self.drv) - 1 - (len(self.drv) - 1) % batch
                self.batches.append((list(range(k0, len(self.drv))), api.NativeBatch(self.drv[k0:])))
        self.frames = self.refsum = self.keys = 0
        # GOP steady state, untimed and independent of --warmup: every chunk past two altref periods, phases staggered so
        # that every step sees the long-run mix of LAST / LAST+GOLDEN / LAST+GOLDEN+ALTREF frames (and, with a finite GOP, the
        # chunks spread evenly over the positions of the GOP: key frames come one chunk at a time, not all at once)
        pre = [PREROLL + k % ALTREF_RANGE + ((k * gop) // G if gop else 0) for k in range(G)]
        if self.batches:
            for members, nb in self.batches:
                for r in range(max(pre[k] for k in members)):
                    self.step_group(members, nb, [r < pre[k] for k in members])
        else:
            for k in range(G):
                for _ in range(pre[k]):
                    self.step_one(k)
        api.device_synchronize(device)
        self.frames = self.refsum = self.keys = 0

    def step_group(self, members, nb, on=None):
        keys = nb.encode_frame_device([self.ptrs[self.t[k] % self.nd] for k in members], on)
        for i, k in enumerate(members):
            if on is not None and not on[i]:
                continue
            self.t[k] += 1
            self.frames += 1
            if keys[i]:
                self.keys += 1
                continue
            st = self.drv[k].stats()
            self.refsum += 1 + st.last_use_golden + st.last_use_altref

    def step_one(self, k):
        d = self.drv[k]
        key = d.encode_frame_device(*self.ptrs[self.t[k] % self.nd])
        self.t[k] += 1
        self.frames += 1
        if key:
            self.keys += 1
            return
        st = d.stats()
        self.refsum += 1 + st.last_use_golden + st.last_use_altref

    def step(self):
        if self.batches and os.environ.get("VP8_BENCH_PYSTEP"):      # A/B: the groups advanced one by one from Python, in a fixed order
            for members, nb in self.batches:
                self.step_group(members, nb)
            return
        if self.batches:
            # one frame on every group, ONE call: with check_SSIM in the loop a group's next frame needs the verdict on its previous
            # one (a few words the device writes to host memory), and the groups are served as those come in, natively
            # (vp8drv_batches_encode_frame_device); the references searched are read from the drivers' counters afterwards
            keys = self.api.NativeBatch.encode_frame_device_all([nb for _, nb in self.batches],
                                                                [[self.ptrs[self.t[k] % self.nd] for k in members] for members, _ in self.batches])
            for (members, _), kk in zip(self.batches, keys):
                for i, k in enumerate(members):
                    self.t[k] += 1
                    self.keys += int(kk[i])
            self.frames += self.G
            return
        for k in range(self.G):
            self.step_one(k)

    def pin_host_frames(self):
        """the nd source frames once more in page-locked HOST memory (the from_host_memory legs: every frame crosses the link on its way in)"""
        if self.host_ptrs is None:
            import numpy as np
            # (a frame's planes end to end, as a Y4M reader or a decoder holds an I420 frame: one copy per frame)
            self.pinned = [(self.api.HostBuffer(np.concatenate([np.ascontiguousarray(p).reshape(-1) for p in f]), self.device),) for f in self.source]
            self.host_ptrs = [(b[0].data_ptr(), b[0].data_ptr() + f[0].size, b[0].data_ptr() + f[0].size + f[1].size) for b, f in zip(self.pinned, self.source)]
        return self.host_ptrs

    def refs_searched(self):
        return sum(d.stats().refs_searched for d in self.drv)

    def profile(self, kernels):
        for d in self.drv:
            d.hip.profile_enable(kernels)

    def profile_read(self):
        prof = {}
        for d in self.drv:
            for k, (ms, n) in d.hip.profile_read().items():
                pm, pn = prof.get(k, (0.0, 0))
                prof[k] = (pm + ms, pn + n)
        return prof

    def clock_read(self):
        """the loop filter by the kernel's own clock, all chunks: (ms, launches, shader clock GHz) since the last call"""
        tot, n, ghz = 0.0, 0, 0.0
        self.context_switches = 0
        for d in self.drv:
            ms, k, g = d.hip.profile_read_clock()
            tot += ms
            n += k
            ghz += g * k
            self.context_switches += d.hip.profile_context_switches()
        # k_search2 by its own clock: (ms, launches); a batched launch is stamped once, on the batch's first member
        self.s2_clock = [sum(v) for v in zip(*[d.hip.profile_read_search2_clock() for d in self.drv])]
        return tot, n, ghz / max(n, 1)

    def run(self, steps, barrier=None, host=False):
        """time `steps` steps; returns (seconds, host enqueue seconds, refs per frame).  host: the frames come from host memory (batches only)"""
        sync = barrier or (lambda: self.api.device_synchronize(self.device))
        self.frames = self.refsum = self.keys = 0
        refs0 = self.refs_searched()
        sync()
        t0 = time.perf_counter()
        if self.batches and not os.environ.get("VP8_BENCH_ONE_THREAD"):
            # One host thread per group, each advancing its group by `steps` frames (vp8drv_batches_encode_frames_device starts and
            # joins them): with check_SSIM in the loop a group's next frame waits for the verdict on its previous one, and a single
            # thread that serves the groups in turn couples them -- a late verdict holds up seven other streams (same box, M MB/s:
            # one thread 55.3-58.9, a thread per group 60.3-60.4, check_SSIM off 60.7-60.9).
            keys = self.api.NativeBatch.encode_frames_device_all([nb for _, nb in self.batches], steps, self.pin_host_frames() if host else self.ptrs,
                                                                 [[self.t[k] for k in members] for members, _ in self.batches], host=host)
            for (members, _), kk in zip(self.batches, keys):
                for i, k in enumerate(members):
                    self.t[k] += steps
                    self.keys += kk[i]
            self.frames += steps * self.G
        else:
            for _ in range(steps):
                self.step()
        enq = time.perf_counter() - t0
        sync()
        el = time.perf_counter() - t0
        for d in self.drv:
            d.hip.synchronize()   # raises if a bounded device-side wait (loop filter / intra wavefronts) expired: no number then
        if self.batches:
            self.refsum = self.refs_searched() - refs0
        return el, enq, self.refsum / max(self.frames - self.keys, 1)

    def close(self):
        for _, nb in self.batches:
            nb.close()
        self.batches = []
        for d in self.drv:
            d.close()
        self.drv = []
        for f in self.dev_frames + self.pinned:
            for p in f:
                p.free()
        self.dev_frames, self.pinned = [], []

    def oracle_check(self):
        """EVERY chunk's filtered reconstruction, as it stands now, against the CPU oracle loop: a chunk is a closed GOP that started
        with its key frame at frame `phase` of the eight-frame cycle and has coded n frames since, and the committed table
        tests/golden/full_length/chunks_<geometry>.json holds the oracle loop's reconstruction CRCs for every (phase, n) up to its
        length.  None where no table applies (other seed than rank 0's, SSIM target, finite GOP, conformant stream)."""
        import zlib
        W0, H0 = self.source_size
        name = f"chunks_{W0}x{H0}" + ("_last_only" if self.refs == "last" else "")
        doc = golden_digest(name)
        if doc is None or doc.get("seed") != self.seed0 or doc.get("distinct_frames") != self.nd or self.ssim_target != -1.0 or self.gop or self.conformant or not CHECK_SSIM or doc.get("refs", "all") != self.refs:
            return None
        checked, wrong, beyond = 0, [], 0
        for k, d in enumerate(self.drv):
            d.resolve()
            n, phase = self.t[k] - self.t_key[k], self.t_key[k] % self.nd
            if n > doc["frames"]:
                beyond += 1
                continue
            got = [zlib.crc32(p.tobytes()) for p in d.hip.download_last()]
            checked += 1
            if got != doc["recon_crc32"][phase][n - 1]:
                wrong.append(k)
        return {"table": f"tests/golden/full_length/{name}.json", "chunks_checked": checked, "chunks_beyond_the_table": beyond, "differing_chunks": wrong,
                "identical": not wrong, "what": "the filtered reconstruction every chunk stands on, CRC-32 of Y, U, V, against the CPU oracle loop run "
                "over the same frames from the chunk's key frame (scripts/full_length_oracle.py --oracle); the run aborts on a mismatch"}

    def replay_chunk(self, k):
        """Chunk k coded AGAIN, from its key frame to where it stands now, on a fresh driver of its own -- no batch, no other chunk
        beside it, one frame at a time: the filtered reconstruction it ends with must be the chunk's, byte for byte (a closed GOP
        depends on nothing but its own frames).  The self-check of a bench line: outside every timed region."""
        import zlib
        d = self.api.NativeDriver(self.W, self.H, device=self.device, gop_size=self.gop or (1 << 30), altref_range=ALTREF_RANGE, qi_min=0, qi_max=48,
                                  ssim_target=self.ssim_target, device_params=1, check_ssim=CHECK_SSIM, ref_mask=3 if self.refs == "all" else 0,
                                  overlap_filter=0, conformant_stream=self.conformant, **self.src_kw)
        for t in range(self.t_key[k], self.t[k]):
            d.encode_frame_device(*self.ptrs[t % self.nd])
        d.resolve()
        crc = lambda planes: [zlib.crc32(p.tobytes()) for p in planes]
        self.drv[k].resolve()
        a, b = crc(self.drv[k].hip.download_last()), crc(d.hip.download_last())
        sa, sb = self.drv[k].stats(), d.stats()
        d.close()
        return {"chunk": k, "frames_recoded": self.t[k] - self.t_key[k], "crc32_yuv_batched": a, "crc32_yuv_alone": b,
                "key_frames": [sa.key_frames, sb.key_frames], "identical": a == b and sa.key_frames == sb.key_frames}


def side_leg(api, W0, H0, G, refs, ssim_target, steps, warm, device, nd=4, seed=1, batch=1, gop=None, conformant=0):
    # one chunk = one video coded frame after frame: the loop filter on its own stream, GOLDEN/ALTREF searched beside it
    leg = Leg(api, W0, H0, G, refs, ssim_target, nd, device, seed, overlap_filter=1 if G == 1 else 0, batch=batch if G > 1 else 1, gop=gop,
              conformant=conformant)
    for _ in range(warm):
        leg.step()
    leg.clock_read()
    el, enq, nrefs = leg.run(steps)
    lf_ms, lf_n, ghz = leg.clock_read()
    frames = steps * G
    oc = leg.oracle_check()
    if oc and not oc["identical"] and not api.load_library().vp8hip_experiments_compiled_in():
        raise SystemExit(f"bench.py: self-check of the {W0}x{H0} leg FAILED against the oracle digests: {oc}")
    out = {"workload": f"{W0}x{H0}, {'LAST+GOLDEN+ALTREF' if refs == 'all' else 'LAST only'}, SSIM target {ssim_target}, {G} GOP chunk(s) in flight"
                       + (f" in batches of {batch}" if G > 1 and batch > 1 else ""),
           "value": round(leg.mbs * frames / el, 1), "unit": "macroblocks/s", "ms_per_frame": round(el / frames * 1e3, 4),
           "fps": round(frames / el, 1), "frames": frames, "refs_per_frame": round(nrefs, 2), "macroblocks_per_frame": leg.mbs,
           "loop_filter_ms_by_its_own_clock": round(lf_ms / max(lf_n, 1), 4), "shader_clock_ghz": round(ghz, 3),
           "waves_context_switched": leg.context_switches, "self_check_against_the_oracle": oc}
    if gop:
        out["gop_size"], out["key_frames"] = gop, leg.keys
    leg.close()
    return out


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


def literal_gops(api, W0, H0, chunks, gop_len, device, nd, refs="all", bitstream=False, seed=1, frames_out=None, frame_base=0, start=None):
    """`chunks` closed GOPs of `gop_len` frames each on this GPU, each ONE video coded frame after frame from its key frame on (loop
    filter on the chunk's second stream), one host thread per chunk, every frame counted: a BASELINE config as it is written, not
    the saturated steady state of `value`.  bitstream: every frame is also delivered as bytes (vp8drv_get_frame) into
    frames_out[frame_base + chunk * gop_len + t].  Returns (seconds, frames, key frames, frames recoded as key, bytes)."""
    import threading
    leg = Leg(api, W0, H0, 0, refs, -1.0, nd, device, seed)       # the synthetic frames in HBM; no drivers yet
    src = dict(src_width=W0, src_height=H0) if tuple(leg.source_size) != (leg.W, leg.H) else {}
    drv = [api.NativeDriver(leg.W, leg.H, device=device, gop_size=1 << 30, altref_range=ALTREF_RANGE, qi_min=0, qi_max=48, ssim_target=-1.0,
                            device_params=1, check_ssim=CHECK_SSIM, ref_mask=3 if refs == "all" else 0, overlap_filter=1, **src) for _ in range(chunks)]
    if bitstream:      # the entropy stage's scratch: not inside the timed region (the reference allocates everything in init_all)
        for d in drv:
            d.hip.reserve_frame_path_dense()
    keys, nbytes = [0] * chunks, [0] * chunks
    pipelined = bitstream and not os.environ.get("VP8_BENCH_NO_FRAME_PIPELINE")
    native_loop = pipelined and not os.environ.get("VP8_BENCH_PY_VIDEO_LOOP")
    # where the frames land: host memory allocated and touched before the clock starts (the reference's output buffers are init_all()'s)
    video_out = [d.video_out_buffer(gop_len) for d in drv] if native_loop else None

    def work(k):
        d = drv[k]
        pending = None
        if native_loop:
            # the whole loop natively (vp8drv_encode_video_device: encode(t), frame t - 1's bytes, frame t's stage, frame t's verdict):
            # the host's reaction times are on the path -- the later a frame's stage is enqueued behind its verdict, the further it
            # reaches under the next frame's LAST search
            fr, kk = d.encode_video_device(gop_len, leg.ptrs, start=3 * k, out=video_out[k], views=True)
            keys[k] += kk
            nbytes[k] += sum(len(b) for b in fr)
            if frames_out is not None:
                for t, b in enumerate(fr):
                    frames_out[frame_base + k * gop_len + t] = b
            d.hip.synchronize()
            return
        if not bitstream and not os.environ.get("VP8_BENCH_PY_VIDEO_LOOP"):
            d.encode_video_device_no_frames(gop_len, leg.ptrs, start=3 * k)     # (the same calls from C: no interpreter lock between two videos' threads)
            d.hip.synchronize()
            return
        for t in range(gop_len):
            d.encode_frame_device(*leg.ptrs[(3 * k + t) % leg.nd])
            if pipelined:
                # frame t is under way; NOW take frame t - 1's bytes (its entropy stage ran on the context's third stream beside
                # frame t - 1's loop filter and frame t's side work), then enqueue frame t's stage
                if pending is not None:
                    b = d.get_frame_end()
                    nbytes[k] += len(b)
                    if frames_out is not None:
                        frames_out[frame_base + k * gop_len + pending] = b
                d.get_frame_begin()
                pending = t
                keys[k] += int(bool(d.resolve()))
                continue
            if bitstream:
                b = d.get_frame()
                nbytes[k] += len(b)
                if frames_out is not None:
                    frames_out[frame_base + k * gop_len + t] = b
            keys[k] += int(bool(d.resolve())) if (bitstream or t == gop_len - 1) else 0
        if pending is not None:
            b = d.get_frame_end()
            nbytes[k] += len(b)
            if frames_out is not None:
                frames_out[frame_base + k * gop_len + pending] = b
        d.hip.synchronize()

    if start is not None:
        start()             # (all ranks begin their frame loops together)
    api.device_synchronize(device)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(k,)) for k in range(chunks)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    api.device_synchronize(device)
    el = time.perf_counter() - t0
    st = [d.stats() for d in drv]
    out = (el, chunks * gop_len, sum(s.key_frames for s in st), sum(s.redone_as_key for s in st), sum(nbytes), leg.mbs)
    for d in drv:
        d.close()
    leg.close()
    return out


def few_stream_legs(args, api, dist, rank, world, local, nd, barrier, emit=None, which="few"):
    """config5_literal (every N), ref_shard, and at N = 1 config3_literal, single_stream and other_configs: the legs that are one or two
    videos coded frame after frame, and the other geometries.  Run in a fresh process (see main()).  Returns the dict for the JSON line
    on rank 0; `emit` (if given) is also handed every finished leg at once, so that a leg that dies takes only itself along."""
    class _Out(dict):
        def __setitem__(self, k, v):
            dict.__setitem__(self, k, v)
            if emit is not None:
                emit({k: v})
    out = _Out()
    few, other = which == "few", which == "other"
    if rank == 0 and few:
        # every kernel of the path ALONE on the part: one chunk, one stream, nothing beside it, each launch timed by its own dispatch
        # (HIP events) -- the launch durations the roofline fractions are made of (with 48 chunks in flight a launch shares the part)
        solo = Leg(api, args.width, args.height, 1, args.refs, args.ssim_target, nd, local, seed=1)
        solo.profile(api.K_NAMES)
        for _ in range(4):
            solo.step()
        solo.profile_read()
        el_s, _, nrefs_s = solo.run(40)
        out["solo_kernels"] = {"refs_per_frame": round(nrefs_s, 3), "ms_per_launch": {k: round(ms / n, 5) for k, (ms, n) in solo.profile_read().items() if n},
                               "what": "one GOP chunk on one stream, every kernel timed by its own dispatch; nothing else on the part"}
        solo.close()
    if rank == 0 and world == 1 and few:
        # BASELINE configs[2] as it is written: 300 frames, the reference's -g 150 -> two closed GOPs of 150 frames, both in flight,
        # each one video coded frame after frame from its key frame on; every frame counted (2 key frames among the 300)
        el3, n3, k3, r3, _, mbs3 = literal_gops(api, args.width, args.height, 2, 150, local, nd)
        out["config3_literal"] = {"workload": f"{args.width}x{args.height}, 300 frames, -g 150: two closed GOPs of 150 frames in flight on one GPU, LAST+GOLDEN+ALTREF, "
                                              "check_SSIM in the loop, loop filter on the GPU, every frame counted",
                                  "value": round(mbs3 * n3 / el3, 1), "unit": "macroblocks/s", "fps": round(n3 / el3, 1), "ms_per_frame": round(el3 / n3 * 1e3, 4),
                                  "seconds": round(el3, 4), "frames": n3, "key_frames": k3, "frames_redone_as_key": r3}
        s1 = max(200, args.steps)
        ss = side_leg(api, args.width, args.height, 1, args.refs, args.ssim_target, s1, 20, local, nd=nd)
        ss["what"] = "ONE closed GOP coded frame after frame (what configs[2] literally is): bound by the latency of the frame's dependency chain"
        out["single_stream"] = ss
    if few:
        # BASELINE configs[4] as it is written, at every N: 300 frames per GPU = ONE closed GOP of 300 frames on each rank (2400 / 8),
        # coded end to end from its key frame with finished VP8 frames out, the frames gathered to rank 0 over RCCL in frame order
        # (gop_shard.gather_frames); the time includes the gather.  All ranks take part (collective calls).
        from vp8oclenc_amd import gop_shard
        GOP5 = int(os.environ.get("VP8_BENCH_GOP5", "300"))
        local_frames = {}
        el5, n5, k5, r5, b5, mbs5 = literal_gops(api, args.width, args.height, 1, GOP5, local, nd, bitstream=True, seed=1 + rank,
                                                 frames_out=local_frames, frame_base=rank * GOP5, start=barrier)
        # (literal_gops times its frame loop between synchronisations of its own; the clock goes on with the gather.  What is NOT in the
        # time: making the synthetic frames and creating the encoder, which is init_all() in the reference)
        t0 = time.perf_counter()
        gathered = gop_shard.gather_frames(local_frames, GOP5 * world, dist)
        barrier()
        t5 = el5 + (time.perf_counter() - t0)
        if dist is not None:
            t5 = dist.max(t5)
        # outside the time: THIS rank's 300 frames against the CPU oracle loop's (tests/golden/full_length/config5_rank<r>.json: CRC-32
        # and length of every frame, made by scripts/full_length_oracle.py --oracle from the same synthetic frames)
        import zlib
        import numpy as np
        doc = golden_digest(f"config5_rank{rank}") if (args.width, args.height, GOP5) == (1920, 1080, 300) and CHECK_SSIM else None
        mine = [local_frames[rank * GOP5 + t] for t in range(GOP5)]
        differing = -1 if doc is None else sum((zlib.crc32(b), len(b)) != (doc["frame_crc32"][t], doc["frame_len"][t]) for t, b in enumerate(mine))
        verdicts = np.array([differing], np.int64) if dist is None else dist.all_gather(np.array([differing], np.int64)).reshape(-1)
        oracle5 = {"ranks_checked": [r for r in range(world) if verdicts[r] >= 0], "ranks_without_a_committed_digest": [r for r in range(world) if verdicts[r] < 0],
                   "frames_per_rank": GOP5, "differing_frames": int(sum(max(int(v), 0) for v in verdicts)), "identical": not any(int(v) > 0 for v in verdicts),
                   "what": "every frame a rank delivered (CRC-32 + length) against the CPU oracle loop over the same frames, tests/golden/full_length/config5_rank<r>.json"}
        if not oracle5["identical"] and not api.load_library().vp8hip_experiments_compiled_in():
            raise SystemExit(f"bench.py: config5_literal FAILED its check against the oracle digests: {oracle5}")
        if rank == 0:
            assert gathered is not None and len(gathered) == GOP5 * world and all(gathered)
            out["config5_literal"] = {
                "workload": f"{args.width}x{args.height}, {GOP5 * world} frames = one closed GOP of {GOP5} frames on each of {world} GPU(s), LAST+GOLDEN+ALTREF, check_SSIM in the loop, "
                            "finished VP8 frames gathered to rank 0 in frame order (vp8hip_group_gather_bytes: ncclSend / ncclRecv inside the library); the time includes the gather",
                "value": round(mbs5 * GOP5 * world / t5, 1), "unit": "macroblocks/s", "fps": round(GOP5 * world / t5, 1), "seconds": round(t5, 4),
                "frames": GOP5 * world, "key_frames": k5 * world, "bytes_gathered": int(sum(len(b) for b in gathered)),
                "encode_seconds_rank0": round(el5, 4), "self_check_against_the_oracle": oracle5, "n_gpus": world, "rccl_ranks": None if dist is None else int(dist.count()),
                "gpu_framework_in_process": "torch" if "torch" in sys.modules else "none"}
        del gathered, local_frames
        # one GOP split BY REFERENCE over up to three ranks: the searches of a frame on different GPUs, vector nets
        # all_gathered, the filtered reconstruction broadcast.  Needs three ranks to mean anything; with fewer the same exchanges
        # are walked by loopback on rank 0 (what they cost on one GPU), the other ranks only keep the collectives company.
        rs = ref_shard_leg(api, dist, args.width, args.height, local, rank, world, int(os.environ.get("VP8_BENCH_REFSHARD_FRAMES", "60")))
        if rank == 0 and rs is not None:
            out["ref_shard"] = rs
    # The other geometries run in a child of their own: the one-video legs above make low-priority streams (the side stream of
    # vp8hip_filter_overlap, the entropy stage's) and RCCL brings queues too -- a process keeps every hardware queue it ever used, the
    # priority classes have queue sets of their own, and past 24 queues per process the part's scheduler rotates them and
    # context-switches running waves (`waves_context_switched` was 20-47 per leg here while these legs shared the few-stream child).
    if rank == 0 and world == 1 and other and not args.only_bitstream:
        G, B = max(1, args.gops_per_gpu), max(1, min(8, args.batch))
        oc = {}
        for name, leg_args, kw in (
                # 4K: sixteen chunks in eight batches of two (same-box: in batches of 4 55.0, of 2 59.7); 720p: batches of four
                # (48 chunks: twelve streams 102.8, eight streams 98.0)
                ("720p_last_only", (1280, 720, G, "last", -1.0, max(20, args.steps // 2), 5), dict(batch=min(B, 4))),
                ("4k_3refs", (3840, 2160, min(G, 16), "all", -1.0, max(10, args.steps // 4), 3), dict(batch=min(B, 2))),
                ("1080p_ssim93", (1920, 1080, G, "all", 0.93, max(20, args.steps // 2), 5), dict(batch=B)),
                # the reference's default GOP of 150: key frames (a raster-order wavefront each, 1.5 ms alone) among the inter frames;
                # value counts every frame
                ("1080p_gop150", (1920, 1080, G, "all", -1.0, max(20, args.steps // 2), 5), dict(batch=B, gop=150)),
                # vp8hip_conformant_stream (NOT the reference's bytes: the format's predictor, so that the stream decodes to the
                # encoder's own reconstruction): what the opt-in costs
                ("1080p_conformant_stream", (1920, 1080, G, "all", -1.0, max(20, args.steps // 2), 5), dict(batch=B, conformant=1))):
            oc[name] = side_leg(api, *leg_args, local, nd=nd, **kw)
            out["other_configs"] = dict(oc)      # (handed on after every geometry)
    return dict(out)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.spawn):
        sys.exit(spawn_ranks(args))      # before `import torch`: the launcher never initialises the GPU
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
        few_stream_legs(args, api, dist, rank, world, local, nd, barrier, emit, which=args.child_legs)
        if dist is not None:
            dist.close()
        sys.stdout.flush()
        os.close(json_fd)
        sys.exit(0)          # (through the ordinary teardown: every context of every leg has been destroyed by now)
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
                child = subprocess.run([sys.executable, "-X", "faulthandler", os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE,   # (a leg that dies says where, on stderr)
                                       timeout=float(os.environ.get("VP8_BENCH_CHILD_TIMEOUT", "420")))
                child_out, child_rc = child_out + child.stdout, child_rc or child.returncode
            except subprocess.TimeoutExpired as e:
                child_out, child_rc = child_out + (e.stdout or b""), "timeout"
        if dist is not None:
            dist.barrier()
    G = max(1, args.gops_per_gpu)
    B = max(1, min(8, args.batch))   # VP8HIP_MAX_BATCH
    free_before = api.device_mem_info(local)[0]
    leg = Leg(api, args.width, args.height, G, args.refs, args.ssim_target, nd, local, seed=1 + rank,
              overlap_filter=int(os.environ.get("VP8_BENCH_OVERLAP", "0")),   # experiment switch: every chunk's filter on a second stream
              batch=B)
    W, H, mbs = leg.W, leg.H, leg.mbs
    api.device_synchronize(local)
    hbm_used = free_before - api.device_mem_info(local)[0]      # contexts (surfaces, nets, coefficient buffers) + the synthetic frames

    # ---- warmup; every kernel of chunk 0 timed to find the dominant one --------------------------------------
    leg.drv[0].hip.profile_enable(api.K_NAMES)
    for d in leg.drv:
        d.hip.profile_search2_clock(True)    # k_search2 stamps its launches during the warm-up only (the stamping costs 1 %)
    leg.clock_read()
    for _ in range(max(args.warmup, 1)):
        leg.step()
    api.device_synchronize(local)
    warm = leg.drv[0].hip.profile_read()
    for d in leg.drv:
        d.hip.profile_search2_clock(False)
    leg.clock_read()
    s2_ms, s2_n = leg.s2_clock
    per_launch = {k: ms / n for k, (ms, n) in warm.items() if n}
    dominant = max((k for k in per_launch if algorithmic_bytes(k, W, H, 1) > 0), key=lambda k: per_launch[k])
    # each timed kernel costs two event packets per launch (timing four kernels on every chunk cost 6 % of the headline in
    # a same-box A/B): the timed region times only the roofline kernel, on every chunk; the other kernels' launch times
    # come from the warm-up steps of chunk 0 above (same steady state, fifteen other chunks in flight)
    timed = api.K_NAMES if args.profile_all else [dominant]
    leg.profile(timed)
    leg.clock_read()   # restart the in-kernel clock sums

    # ---- timed region: exactly --steps steps, barrier + synchronize on both sides, max over ranks -------------
    elapsed, enqueue_s, nrefs_avg = leg.run(args.steps, barrier)
    own_elapsed = elapsed
    per_rank = None
    if dist is not None:
        elapsed = dist.max(elapsed)          # the contract: the slowest rank's time
        import numpy as _np
        per_rank = dist.all_gather(_np.array([own_elapsed, float(nrefs_avg)], _np.float64))     # [world, 2] on every rank
    prof = leg.profile_read()
    clk_ms, clk_n, clk_ghz = leg.clock_read()
    for d in leg.drv:
        d.resolve()
    redone = sum(d.stats().redone_as_key for d in leg.drv)
    # ---- self-check, outside the timed region: ONE chunk of the timed run coded again from its key frame on a driver of its own
    # (no batch, no other chunk in flight); the reconstruction it ends with must be the timed chunk's.  A run whose timed frames
    # are not the frames a single un-batched encoder produces prints no line.
    verify = leg.replay_chunk((7 * (rank + 1)) % G)
    verify["against_the_oracle"] = leg.oracle_check()
    if verify["against_the_oracle"] and not verify["against_the_oracle"]["identical"] and not experiment_build:
        raise SystemExit(f"bench.py: self-check FAILED -- chunks of the timed region do not stand on the reconstruction the CPU oracle loop reaches: {verify['against_the_oracle']}")
    if not verify["identical"] and not experiment_build:      # (an experiment build leaves work out of launches: its line says INVALID)
        raise SystemExit(f"bench.py: self-check FAILED -- the timed region's chunk {verify['chunk']} does not end where the same frames coded alone end: {verify}")
    frames_per_gpu = args.steps * G
    if args.refs == "all" and args.steps * G >= 2 * ALTREF_RANGE and nrefs_avg < 2.7:
        raise SystemExit(f"bench.py: the timed frames averaged {nrefs_avg:.2f} references per frame; LAST+GOLDEN+ALTREF in GOP "
                         "steady state is 2.8 -- this would not be BASELINE configs[2]")

    out = None
    if rank == 0:
        value = mbs * frames_per_gpu * world / elapsed
        ms_frame = elapsed / frames_per_gpu * 1e3
        ms_k, n_k = prof[dominant]
        avg_ms = ms_k / max(n_k, 1)
        items = B if (B > 1 and G > 1) else 1       # a batched launch does the stage for B chunks
        abytes = algorithmic_bytes(dominant, W, H, nrefs_avg) * items
        achieved = abytes / (avg_ms * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dominant, W, H)
        traffic = None if traffic is None else traffic * items   # the PMC pass ran one chunk per launch
        path_bytes = 3000.0          # ~3.0 KB of compulsory HBM traffic per macroblock, whole inter path, 3 references
        roof = {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "traffic_source": traffic_src,
                "basis": "time-shared launches (replaced by the solo launch where the side legs ran: see `solo`)",
                "time_shared": {"avg_launch_ms": round(avg_ms, 5), "algorithmic_bytes_per_launch": int(abytes), "chunks_per_launch": items, "launches": int(n_k),
                                "achieved": round(achieved, 3), "frac": round(achieved / HBM_PEAK_GBS, 6),
                                "launches_in_flight": round(n_k * avg_ms * 1e-3 / elapsed, 2),
                                "what": "the dominant kernel's launches in the timed region by HIP events of their own dispatch, all chunks: with 48 "
                                        "chunks in 8 batches a launch shares the part with the other batches' kernels (launches_in_flight of this "
                                        "kernel alone), so this duration says how long a launch lasts, not how fast the kernel is"},
                "path": {"algorithmic_bytes_per_macroblock": path_bytes, "achieved": round(value * path_bytes / 1e9, 3),
                         "frac": round(value * path_bytes / 1e9 / HBM_PEAK_GBS, 6),
                         "what": "the whole inter path: SURVEY 8(d)'s compulsory bytes per macroblock x macroblocks per second of `value`"},
                "note": "None of this path's kernels is HBM-bound (integer search / transform / a serial filter chain): the binding resource is "
                        "VALU issue, see issue_roofline.  kernel_clock: the time-shared launches by the kernel's own clock"}
        lf_clock = None
        if clk_n:
            kms = clk_ms / clk_n             # every member of a batched launch stamps its own frame: this is per chunk
            cb = algorithmic_bytes("loop_filter", W, H, nrefs_avg)
            lf_clock = {"kernel": "loop_filter", "avg_ms_per_chunk": round(kms, 5), "frames": int(clk_n), "achieved": round(cb / (kms * 1e-3) / 1e9, 3),
                        "frac": round(cb / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                        "how": "s_memrealtime (100 MHz) at the start of the kernel's first band and at the end of its last row",
                        "shader_clock_ghz": round(clk_ghz, 3),
                        "waves_context_switched": leg.context_switches}   # launches whose last wave changed hardware slots: 0 unless the process's queues are oversubscribed
            if dominant == "loop_filter":
                roof["kernel_clock"] = lf_clock
        if dominant == "search2" and s2_n:
            kms = s2_ms / s2_n
            roof["kernel_clock"] = {"kernel": "search2", "avg_launch_ms": round(kms, 5), "launches": int(s2_n), "achieved": round(abytes / (kms * 1e-3) / 1e9, 3),
                                    "frac": round(abytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                                    "how": "s_memrealtime (100 MHz): earliest workgroup start to latest workgroup end of a launch, every 64th workgroup "
                                           "stamping (launch_clock_end, vp8hip_dev.h), over the warm-up steps (same steady state; the stamping costs "
                                           "1 % of throughput and is off in the timed region)"}
        others = {}
        for k, (ms, n) in {**warm, **prof}.items():
            if k == dominant or n == 0:
                continue
            b = algorithmic_bytes(k, W, H, nrefs_avg) * items
            a = b / (ms / n * 1e-3) / 1e9 if b else None
            others[k] = {"avg_launch_ms": round(ms / n, 5), "algorithmic_bytes_per_launch": int(b), "achieved_GBs": None if a is None else round(a, 3),
                         "frac": None if a is None else round(a / HBM_PEAK_GBS, 6)}
        out = {
            "metric": ("INVALID (experiment build: work left out of launches) " if experiment_build else "") + "macroblocks/sec inter-frame (ME+DCT+loopfilter), 1080p",
            "value": round(value, 1),
            "unit": "macroblocks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8/int32", "data": "synthetic",
            "config": {"workload": f"{args.width}x{args.height} YUV420 inter frames, {'LAST+GOLDEN+ALTREF' if args.refs == 'all' else 'LAST only'} "
                                   f"(avg {nrefs_avg:.2f} refs/frame, GOP steady state), loop filter on GPU, {G} GOP chunk(s) in flight per GPU"
                                   + (f", {B} chunks per batched launch ({(G + B - 1) // B} streams)" if B > 1 else ""),
                       "step": f"one inter frame on each of the {G} GOP chunks = {G} frames per GPU",
                       "wrk_size": [W, H], "source_size": list(leg.source_size), "padding": "on the device, inside the step (copy_with_padding)" if tuple(leg.source_size) != (W, H) else "none needed",
                       "macroblocks_per_frame": mbs, "ssim_target": args.ssim_target, "qi_ladder": list(api.quantizer_ladders(0, 48)[0]),
                       "altref_range": ALTREF_RANGE, "preroll_frames_per_chunk": f"{PREROLL}..{PREROLL + ALTREF_RANGE - 1}", "frames_per_gpu": frames_per_gpu,
                       "gops_per_gpu": G, "chunks_per_batched_launch": B, "hbm_bytes_in_use": int(hbm_used), "hbm_bytes_per_chunk": int(hbm_used // G), "refs_per_frame": round(nrefs_avg, 3),
                       "ms_per_frame": round(ms_frame, 5),
                       "segment_params": "device, inside the step", "frame_loop": "native (vp8_driver.cpp), one call per frame",
                       "check_ssim": ("on the device inside the step: intra fallback, filter update at min SSIM > 0.95, verdict read one call later"
                                      if CHECK_SSIM else "OFF (A/B run: not the reference's loop)"),
                       "frames_redone_as_key": redone, "frames_with_filter_update": None,
                       "batch_prep_stream": int(lib.vp8hip_batch_prep_mode()),     # 0 = none (default), 1 = per batch, 2 = one for all
                       "experiment_switches": "COMPILED IN" if experiment_build else "compiled out",
                       "hip_runtime_version": int(lib.vp8hip_runtime_version()), "gpu_framework_in_process": "torch" if "torch" in sys.modules else "none",
                       "process_group": None if dist is None else "vp8hip_group_* (RCCL inside libvp8hip.so; id by file rendezvous)",
                       "cpu_affinity": affinity, "hw_queues": int(lib.vp8hip_hw_queues()), "hw_queues_set_by": "the environment" if os.environ.get("GPU_MAX_HW_QUEUES") else "libvp8hip.so at load time", "launcher": "self-spawned ranks" if os.environ.get("VP8_BENCH_CHILD") else ("torchrun (launcher only)" if "TORCHELASTIC_RUN_ID" in os.environ else "single process")},
            "roofline": roof,
            "loop_filter_by_its_own_clock": lf_clock,
            "issue_roofline": issue_roofline(W, H, nrefs_avg, ms_frame, {**warm, **prof}, clk_ghz if clk_n else None),
            "kernels_ms_per_launch_warmup": {k: round(v, 5) for k, v in sorted(per_launch.items(), key=lambda kv: -kv[1])},
            "other_kernels": others,
            "fps": round(frames_per_gpu * world / elapsed, 2),
            "timed_region_s": round(elapsed, 4),
            # N ranks were seen: RCCL's own count of the communicator, and what every rank measured on its own clock
            "rccl_ranks": None if dist is None else int(dist.count()),
            "per_rank": None if per_rank is None else [{"rank": r, "timed_region_s": round(float(per_rank[r][0]), 4),
                                                        "value": round(mbs * frames_per_gpu / float(per_rank[r][0]), 1),
                                                        "refs_per_frame": round(float(per_rank[r][1]), 3)} for r in range(world)],
            "self_check": dict(verify, what="one chunk of the timed region coded again from its key frame on an un-batched driver of its own: "
                                            "CRC-32 of the final filtered reconstruction (Y, U, V), key frames; and (against_the_oracle) EVERY chunk's "
                                            "reconstruction against the committed digests of the CPU oracle loop; the run aborts on a mismatch"),
            "host_enqueue_ms_per_frame": round(enqueue_s / frames_per_gpu * 1e3, 4),
        }
    leg.profile([])
    # ---- side legs: reported next to the headline value, never as it ---------------------------------------------
    if rank == 0 and world == 1 and not args.no_side_legs:
        # (at least 40 frames per chunk: the leg starts from an idle part with its threads 200 us apart, and over 20 frames that start
        # is 3 % of the rate -- 52.9 against 54.3 M MB/s at 40 and 54.8 at 120 on one box)
        out["with_bitstream"] = bitstream_leg(leg, max(40, args.steps))
        # ... and where the chunks stand after the leg, against the CPU oracle loop again (the frames' bytes are held against an un-batched
        # driver inside the leg; the reconstructions they leave behind against the committed table)
        oc = leg.oracle_check()
        out["with_bitstream"]["self_check_against_the_oracle"] = oc
        if oc and not oc["identical"] and not experiment_build:
            raise SystemExit(f"bench.py: self-check FAILED -- after the frames-out leg chunks do not stand on the oracle loop's reconstruction: {oc}")
        if leg.batches and not os.environ.get("VP8_BENCH_PY_BITSTREAM"):
            # the same two loops with the host-device link in them: every source frame copied in from page-locked host memory inside the
            # timed loop (vp8hip_batch_upload_current: a copy stream per batch, two staging buffers per member), first without, then with
            # the finished frames going back; the frames' bytes are held against an un-batched driver fed from DEVICE memory, the
            # reconstructions against the oracle's table
            hs = 40      # (with the legs before it the driver's command stays inside the oracle's table of 176 frames per chunk: 166)
            leg.pin_host_frames()
            leg.run(2, host=True)        # untimed: the batches make their copy streams and staging buffers on first use
            el_h, _, _ = leg.run(hs, host=True)
            oc_in = leg.oracle_check()
            both = bitstream_leg(leg, hs, host=True)
            oc_both = leg.oracle_check()
            src_bytes = leg.source_size[0] * leg.source_size[1] * 3 // 2
            out["from_host_memory"] = {
                "frames_in": {"value": round(leg.mbs * hs * leg.G / el_h, 1), "unit": "macroblocks/s", "fps": round(hs * leg.G / el_h, 1),
                              "host_to_device_GBs": round(src_bytes * hs * leg.G / el_h / 1e9, 2), "self_check_against_the_oracle": oc_in},
                "frames_in_and_out": {"value": both["value"], "unit": "macroblocks/s", "fps": both["fps"], "avg_frame_bytes": both["avg_frame_bytes"],
                                      "host_to_device_GBs": round(src_bytes * both["fps"] / 1e9, 2), "self_check": both["self_check"],
                                      "self_check_against_the_oracle": oc_both},
                "source_bytes_per_frame": src_bytes, "frames": hs * leg.G,
                "what": "the headline's chunks and native loop with every source frame copied in from page-locked host memory inside the timed region "
                        "(vp8hip_batch_upload_current, the reference's clEnqueueWriteBuffer at vp8enc.cpp:386-388), and with the finished frames "
                        "delivered to host memory as well: the whole-job rates WITH the host-device link in them.  Never `value`."}
            for o in (oc_in, oc_both):
                if o and not o["identical"] and not experiment_build:
                    raise SystemExit(f"bench.py: self-check FAILED -- after a from_host_memory leg chunks do not stand on the oracle loop's reconstruction: {o}")
    host_frames = leg.host_frames
    if run_children:
        if rank == 0:
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
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        os.sched_setaffinity(0, all_cpus)       # the CPU baseline gets every host core again, not the GPU's NUMA node only
        out["cpu_baseline"] = cpu_baseline(args, api, host_frames, W, H, mbs)
    leg.close()              # every context destroyed (vp8drv_batch_destroy, vp8drv_destroy -> vp8hip_destroy): the exit code is real
    if dist is not None:
        dist.close()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    sys.stderr.flush()


def ref_shard_leg(api, dist, W0, H0, local, rank, world, nframes):
    """ONE video with a frame's reference searches spread over min(world, 3) GPUs (vp8oclenc_amd/ref_shard.py; the exchanges are the
    library's: vp8hip_shard_share_search / vp8hip_shard_share_last, RCCL on the context's stream, no host synchronisation per
    frame): ms per frame.  With one rank the communicator has one member (what the calls cost on one GPU)."""
    try:
        from vp8oclenc_amd import ref_shard
        from vp8oclenc_amd.synth import SynthSequence
        members = min(world, 3)
        uid = [ref_shard.shard_unique_id() if rank == 0 else None]
        if dist is not None and world > 1:
            uid = [dist.broadcast_bytes(uid[0], ref_shard.SHARD_ID_BYTES, root=0)]     # (every rank takes part; the first three form the communicator)
        if rank >= members:
            return None
        seq = SynthSequence(W0, H0, seed=9)
        frames = [tuple(api.to_device(p, local) for p in seq.frame(t)) for t in range(6)]     # resident in HBM, like every leg's frames
        be = ref_shard.HipRefBackend(seq.W, seq.H, device=local)
        be.shard_init(uid[0], rank, members)
        drv = ref_shard.RefShardDriver(be, None, seq.W, seq.H, altref_range=ALTREF_RANGE, download=False, device_segments=True)
        for t in range(4):
            drv.encode_frame(*frames[t % len(frames)])
        be.synchronize()
        be.shard_max(0.0)          # (barrier)
        t0 = time.perf_counter()
        for t in range(4, 4 + nframes):
            drv.encode_frame(*frames[t % len(frames)])
        be.synchronize()
        el = be.shard_max(time.perf_counter() - t0)
        mbs = (seq.W // 16) * (seq.H // 16)
        out = {"workload": f"{W0}x{H0}, one GOP, a frame's LAST / GOLDEN / ALTREF searches on " + (f"{members} GPUs" if members > 1 else "one GPU (a communicator of one rank)"),
               "value": round(mbs * nframes / el, 1), "unit": "macroblocks/s", "ms_per_frame": round(el / nframes * 1e3, 4), "frames": nframes,
               "bytes_of_nets_shared_per_frame": int(drv.bytes_gathered / (nframes + 4)), "bytes_broadcast_per_frame": int(drv.bytes_broadcast / (nframes + 4)),
               "ranks": members,
               "what": "vp8hip_inter_search on every rank's references, vp8hip_shard_share_search (one group of RCCL broadcasts, in place in the "
                       "context's nets), vp8hip_inter_finish + loop filter on rank 0, vp8hip_shard_share_last (the padded planes out of rank 0's "
                       "frame pool into the others'): all on the context's stream, no host synchronisation per frame; frames resident in HBM, "
                       "segment data on the device"}
        be.close()
        return out
    except Exception as e:      # a side leg is a report, never a reason to lose the bench line
        return {"error": repr(e)[:300]}


def issue_roofline(W, H, nrefs, ms_frame, prof, held_clock_ghz=None):
    """VALU issue CYCLES per frame against the chip's capacity (256 CUs x 4 SIMDs x 2.4 GHz SIMD-cycles per second).
    profiles/pmc_valu.json: wave64 instructions per launch by opcode class from the committed rocprofv3 --pmc pass and the
    disassembly, and the measured issue cost of each class (scripts/ubench/valu_rates.hip).  Two peaks are quoted: the
    guide's 2 cycles per wave64 VALU instruction (1 229 wave-instr/ns chip-wide) and what this instruction mix can reach
    at its measured per-opcode costs."""
    t = _profile_json("pmc_valu.json")
    g = (t or {}).get(f"{W}x{H}")
    if not g or "cycle_model" not in (t or {}):
        return None
    cm = t["cycle_model"]
    simd_cycles_per_ns = cm["simds"] * cm["clock_ghz"]
    path_keys = [k for k in g if k not in ("_meta", "loop_filter4")]     # (batches launch the loop filter's form 3; form 4 is the one-video kernel)
    insts = {k: (g[k]["per_ref"] * nrefs if "per_ref" in g[k] else g[k]["fixed"]) for k in path_keys}
    cycles = {k: (g[k].get("cycles_per_ref", 0) * nrefs if "per_ref" in g[k] else g[k].get("cycles_fixed", 0)) for k in path_keys}
    tot_i, tot_c = sum(insts.values()), sum(cycles.values())
    ns = ms_frame * 1e6
    out = {"bound": "valu_issue", "unit": "SIMD issue cycles", "peak_simd_cycles_per_ns": simd_cycles_per_ns,
           "path": {"instructions_per_frame": int(tot_i), "issue_cycles_per_frame": int(tot_c),
                    "frac_of_issue_cycles": round(tot_c / (ns * simd_cycles_per_ns), 4),
                    "wave_instr_per_ns": round(tot_i / ns, 1), "frac_of_2cycle_peak": round(tot_i * 2 / (ns * simd_cycles_per_ns), 4),
                    "shader_clock_held_ghz": None if not held_clock_ghz else round(held_clock_ghz, 3),
                    "frac_of_issue_cycles_at_held_clock": None if not held_clock_ghz else round(tot_c / (ns * cm["simds"] * held_clock_ghz), 4)},
           "source": t.get("source"), "cost_source": cm.get("source"), "kernels": {}}
    for k in ("search2", "search1_l0", "mb"):
        if k in prof and prof[k][1] and k in insts:
            kns = prof[k][0] / prof[k][1] * 1e6
            out["kernels"][k] = {"instructions_per_launch": int(insts[k]), "issue_cycles_per_launch": int(cycles[k]),
                                 "avg_launch_ms": round(kns * 1e-6, 5), "frac_of_issue_cycles": round(cycles[k] / (kns * simd_cycles_per_ns), 4),
                                 "frac_of_2cycle_peak": round(insts[k] * 2 / (kns * simd_cycles_per_ns), 4),
                                 "note": "launch time measured with all chunks in flight: other chunks' waves share the SIMDs"}
    return out


def bitstream_leg(leg, nb, host=False):
    """the same chunks with finished VP8 frames delivered to host memory (vp8drv_get_frame: the whole entropy stage on the
    device), one host thread per GOP chunk.  host: the source frames come from page-locked host memory as well (native loop only)"""
    import threading
    G = leg.G
    nbytes = [0] * G
    checks, t_before = [None] * G, None

    def worker(k):
        d = leg.drv[k]
        for _ in range(nb):
            d.encode_frame_device(*leg.ptrs[leg.t[k] % leg.nd])
            leg.t[k] += 1
            nbytes[k] += len(d.get_frame())

    def group_worker(members, batch):     # batched launches: the group's frames in one call, then every member's bytes
        for _ in range(nb):
            leg.step_group(members, batch)
            if os.environ.get("VP8_BENCH_ENT_BATCH", "1") != "0":
                batch.get_frames_begin()
            else:
                for k in members:
                    leg.drv[k].get_frame_begin()
            for k in members:
                nbytes[k] += len(leg.drv[k].get_frame_end())

    for k in range(G):   # untimed: the entropy stage allocates its scratch on first use (sized for the densest frame: the native loop
        leg.drv[k].hip.reserve_frame_path_dense()      # starts frame t + 1 before it takes frame t's bytes, so no frame may need a second coding)
        for _ in range(2):
            leg.drv[k].encode_frame_device(*leg.ptrs[leg.t[k] % leg.nd])
            leg.t[k] += 1
            leg.drv[k].get_frame()
    leg.api.device_synchronize(leg.device)
    tb = time.perf_counter()
    if leg.batches and not os.environ.get("VP8_BENCH_PY_BITSTREAM"):
        # the native loop: a host thread per batch, every frame coded and delivered (vp8drv_batches_encode_frames_device with bytes_out)
        t_before = list(leg.t)
        _, nbo, chk = leg.api.NativeBatch.encode_frames_device_all([b for _, b in leg.batches], nb, leg.pin_host_frames() if host else leg.ptrs,
                                                                   [[leg.t[k] for k in m] for m, _ in leg.batches], frames_out="check", host=host)
        for (members, _), row, crow in zip(leg.batches, nbo, chk):
            for i, k in enumerate(members):
                leg.t[k] += nb
                nbytes[k] += row[i]
                checks[k] = crow[i]
        th = leg.batches
    else:
        th = ([threading.Thread(target=group_worker, args=(m, b)) for m, b in leg.batches] if leg.batches
              else [threading.Thread(target=worker, args=(k,)) for k in range(G)])
        for t in th:
            t.start()
        for t in th:
            t.join()
    leg.api.device_synchronize(leg.device)
    eb = time.perf_counter() - tb
    self_check = None
    if t_before is not None:
        # self-check, outside the timed region: chunk k coded again from its key frame on a driver of its own, one frame at a time; the
        # frames it delivers over the leg's span must be the leg's, byte for byte (vp8drv_frame_check folded over every frame)
        k = 5 % G
        d = leg.api.NativeDriver(leg.W, leg.H, device=leg.device, gop_size=leg.gop or (1 << 30), altref_range=ALTREF_RANGE, qi_min=0, qi_max=48,
                                 ssim_target=leg.ssim_target, device_params=1, check_ssim=CHECK_SSIM, ref_mask=3 if leg.refs == "all" else 0,
                                 conformant_stream=leg.conformant, **leg.src_kw)
        h = size = 0
        for t in range(leg.t_key[k], leg.t[k]):
            d.encode_frame_device(*leg.ptrs[t % leg.nd])
            if t >= t_before[k]:
                f = d.get_frame()
                h, size = leg.api.frame_check(h, f), size + len(f)
        d.close()
        self_check = {"chunk": k, "frames": nb, "bytes": [int(nbytes[k]), int(size)], "frame_check": [int(checks[k]), int(h)],
                      "identical": int(checks[k]) == int(h) and int(nbytes[k]) == int(size),
                      "what": "the leg's frames of one chunk against the same frames delivered by an un-batched driver of its own (every byte, in order)"}
        if not self_check["identical"] and not leg.api.load_library().vp8hip_experiments_compiled_in():     # (an experiment build leaves launches out: its line says INVALID)
            raise SystemExit(f"bench.py: self-check of the frames-out leg FAILED: {self_check}")
    return {"value": round(leg.mbs * nb * G / eb, 1), "unit": "macroblocks/s", "fps": round(nb * G / eb, 1), "frames": nb * G,
            "host_threads_per_gpu": len(th), "avg_frame_bytes": int(sum(nbytes) / (nb * G)), "self_check": self_check,
            "what": "native frame loop + vp8drv_get_frame: coefficient partitions and first partition coded on the device, finished "
                    "frames in host memory (byte-identical to the reference's output)"}


def cpu_baseline(args, api, host_frames, W, H, mbs):
    """Times oracle/vp8_oracle.c (the checker; OpenMP over blocks/MBs) on the host cores: kind 'port'."""
    # all host cores this process may run on (libgomp reads the variable when liboracle.so is loaded)
    os.environ["OMP_NUM_THREADS"] = str(len(os.sched_getaffinity(0)))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    from oracle_lib import Oracle
    lastqi, _ = api.quantizer_ladders(0, 48)        # reference defaults, init.h:1548-1603
    segs = []
    for y, _, _ in host_frames:                     # host parameter producers, outside the timed loop
        red, sharp = api.loopfilter_strength(y)
        segs.append(api.prepare_segments_data(False, lastqi, 0, red, sharp))
    ora = Oracle(W, H, args.ssim_target)
    all_threads = int(Oracle.lib().vp8o_num_threads())
    ora.upload_last(*host_frames[0])
    ora.set_segments(segs[1])
    # warm once with LAST only (sets golden = altref = LAST like the frame after a key frame), then time
    ora.upload_current(*host_frames[1])
    ora.inter_transform(1, 1, 0, 0)
    ora.loop_filter()
    # The restatement is a sequence of short parallel loops with a barrier behind each: on a host with hundreds of hardware threads the
    # barriers cost more than the last doubling of threads brings.  Two frames at each of a few team sizes, the sample at the best one.
    tried = {}
    for n_thr in sorted({all_threads, 128, 64, 32, 16} & set(range(1, all_threads + 1)), reverse=True):
        Oracle.lib().vp8o_set_num_threads(n_thr)
        t0 = time.perf_counter()
        for k in range(2):
            ora.set_segments(segs[(2 + k) % len(segs)])
            ora.upload_current(*host_frames[(2 + k) % len(host_frames)])
            ora.inter_transform(0, 0, 1, 1)
            ora.loop_filter()
        tried[n_thr] = round(mbs * 2 / (time.perf_counter() - t0), 1)
    threads = max(tried, key=tried.get)
    Oracle.lib().vp8o_set_num_threads(threads)
    n, t0 = 0, time.perf_counter()
    while True:
        i = (2 + n) % len(host_frames)
        ora.set_segments(segs[i])
        ora.upload_current(*host_frames[i])
        ora.inter_transform(0, 0, 1, 1)
        ora.loop_filter()
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 64:   # a bounded sample: ~12 s of host time
            break
    out = {"value": round(mbs * n / el, 1), "unit": "macroblocks/s", "cores": threads, "kind": "port",
           "value_per_core": round(mbs * n / el / max(threads, 1), 1), "threads_tried": {str(k): v for k, v in sorted(tried.items())},
           "host_hardware_threads": all_threads,
           "sample": f"{n} inter frames {W}x{H}, 3 references, oracle/vp8_oracle.c with OpenMP on {threads} threads, "
                     f"{el:.1f} s"}
    # the same restatement on ONE thread (what a core does when it does not wait for 255 others at every kernel's barrier)
    if args.cpu_seconds >= 5:
        lib = Oracle.lib()
        lib.vp8o_set_num_threads(1)
        t1 = time.perf_counter()
        ora.set_segments(segs[2 % len(segs)])
        ora.upload_current(*host_frames[2 % len(host_frames)])
        ora.inter_transform(0, 0, 1, 1)
        ora.loop_filter()
        e1 = time.perf_counter() - t1
        lib.vp8o_set_num_threads(all_threads)
        out["port_on_one_thread"] = {"value": round(mbs / e1, 1), "unit": "macroblocks/s", "cores": 1, "kind": "port",
                                     "sample": f"1 inter frame {W}x{H}, 3 references + loop filter, {e1:.1f} s",
                                     "parallel_efficiency_of_the_full_run": round((mbs * n / el) / (mbs / e1) / max(threads, 1), 3)}
    ora.close()
    # beside it, where oracle/_ref travelled: the REFERENCE'S OWN kernels (GPU_kernels.cl + CPU_kernels.cl compiled for x86,
    # oracle/build_ref.sh) through the same frame -- work-item loops on one core, the way oracle/ref_driver.c drives them
    try:
        import numpy as np
        from oracle_lib import ref_stages
        from pipeline import run_inter_frame
        st = ref_stages()
        if st is not None and args.cpu_seconds >= 5:
            t0 = time.perf_counter()
            run_inter_frame(st, host_frames[3 % len(host_frames)], [host_frames[2 % len(host_frames)], host_frames[0], host_frames[1]],
                            np.asarray(segs[3 % len(segs)]).reshape(4, 11), 1, 1, args.ssim_target)
            el1 = time.perf_counter() - t0
            out["reference_kernels_on_one_core"] = {"value": round(mbs / el1, 1), "unit": "macroblocks/s", "cores": 1, "kind": "reference",
                                                    "sample": f"1 inter frame {W}x{H}, 3 references + loop filter, the reference's kernels compiled for x86 "
                                                              f"(oracle/_ref/libvp8ref.so), {el1:.1f} s"}
    except Exception as e:      # the baseline is a report, never a reason to lose the bench line
        out["reference_kernels_on_one_core"] = {"error": repr(e)[:200]}
    return out


if __name__ == "__main__":
    main()
