"""G GOP chunks of one geometry on one GPU (class Leg: the headline's and the side legs' workload) and the self-checks a leg can make:
every chunk's reconstruction against the committed digests of the CPU oracle loop, one chunk coded again on a driver of its own."""
from __future__ import annotations

import os
import time

from .common import ALTREF_RANGE, CHECK_SSIM, PREROLL, golden_digest

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
        with its key frame at frame `phase` of the eight-frame cycle and has coded n frames since, and the committed tables
        tests/golden/full_length/chunks_<geometry>[_<variant>].json hold the oracle loop's reconstruction CRCs for (phase, n) up to their
        length (variants: _last_only, _ssim93 -- the four-pass ladder --, _conformant, _phase0_long -- one video longer than the chunk
        table).  A finite GOP uses the plain table: n frames past a key frame the chunk coded itself it stands where a chunk that STARTED
        with that key frame stands (closed GOPs, intra_part.h:1091-1098).  None where no table applies (other seed than rank 0's, another
        SSIM target).  `identical` is None -- not true -- when not one chunk could be looked up."""
        import zlib
        W0, H0 = self.source_size
        base = f"chunks_{W0}x{H0}" + ("_last_only" if self.refs == "last" else "")
        variant = "_conformant" if self.conformant else ("_ssim93" if abs(self.ssim_target - 0.93) < 1e-6 else "")
        if not CHECK_SSIM or (self.ssim_target != -1.0 and not variant) or (self.conformant and self.ssim_target != -1.0):
            return None
        docs = [(n, d) for n, d in ((n, golden_digest(n)) for n in ([base + variant] + ([base + "_phase0_long"] if not variant else [])))
                if d is not None and d.get("seed") == self.seed0 and d.get("distinct_frames") == self.nd and d.get("refs", "all") == self.refs
                and float(d.get("ssim_target", -1.0)) == float(self.ssim_target) and int(d.get("conformant", 0)) == int(self.conformant)]
        if not docs:
            return None
        checked, wrong, beyond, used = 0, [], 0, set()
        for k, d in enumerate(self.drv):
            d.resolve()
            n, phase = self.t[k] - self.t_key[k], self.t_key[k] % self.nd
            if self.gop:      # the chunk's own key frames: frames since the last one, and the phase that one had
                phase, n = (self.t_key[k] + ((n - 1) // self.gop) * self.gop) % self.nd, (n - 1) % self.gop + 1
            doc = next(((nm, dd) for nm, dd in docs if phase in dd.get("phases", range(self.nd)) and n <= dd["frames"]), None)
            if doc is None:
                beyond += 1
                continue
            got = [zlib.crc32(p.tobytes()) for p in d.hip.download_last()]
            checked += 1
            used.add(doc[0])
            if got != doc[1]["recon_crc32"][phase][n - 1]:
                wrong.append(k)
        return {"tables": [f"tests/golden/full_length/{n}.json" for n in sorted(used)] or [f"tests/golden/full_length/{n}.json" for n, _ in docs],
                "chunks_checked": checked, "chunks_beyond_the_table": beyond, "differing_chunks": wrong,
                "identical": (not wrong) if checked else None,
                "what": "the filtered reconstruction every chunk stands on, CRC-32 of Y, U, V, against the CPU oracle loop run over the same frames from "
                        "the chunk's key frame (scripts/full_length_oracle.py --oracle); the run aborts on a mismatch; identical is null when no chunk "
                        "could be looked up"}

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
    if oc and oc["identical"] is False and not api.load_library().vp8hip_experiments_compiled_in():
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

