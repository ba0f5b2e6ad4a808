"""The product north_star names, timed: the REFERENCE'S OWN main() (src/vp8enc.cpp, built against libvp8hip.so by
oracle/ref_main/build.sh -> oracle/_ref/vp8oclenc_hip*) coding BASELINE configs[2] -- 1920x1080, 300 frames, the reference's defaults
(-g 150, three references, check_SSIM, loop filter) -- from a YUV4MPEG2 file on tmpfs to an IVF file on tmpfs, a fresh process per run.

    vp8oclenc_hip        every stage on the device, the reference's loop with its SYNCHRONOUS calls (vp8hip_upload_current, the host's
                         copy_with_padding / get_loopfilter_strength / scene_change scans, vp8hip_check_ssim, vp8hip_encode_frame)
    vp8oclenc_hip_host   the reference's host intra path, check_SSIM and encode_header kept (-DVP8HIP_KEEP_HOST_STAGES)
    vp8oclenc_hip_fast   the reference's control flow with the library's asynchronous entry points and device-side scans in place of the
                         host scans and blocking calls (-DVP8HIP_FAST; frames leave one iteration late)
    y4m_to_ivf           the product's own loop (scripts/native/y4m_to_ivf.cpp): the comparator, and the bytes every build must write

Per program: frames/s of the frame loop (the program's own clock from the end of init_all() to the start of finalize(): reads, coding,
writes), frames/s of the whole process (exec to exit, with the runtime's start-up and context creation), whether the .ivf is byte for byte
y4m_to_ivf's, and for the last run the host's time per call site.  Needs an MI355X; where oracle/_ref is absent the leg says so.
(oracle/_ref holds test infrastructure built from the reference's sources; this leg executes those binaries as the thing MEASURED NEXT TO the
product -- never as the product, and `value` never comes from here.)
"""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
PROGRAMS = [("reference_main_device_stages", "vp8oclenc_hip"), ("reference_main_host_stages", "vp8oclenc_hip_host"), ("reference_main_fast", "vp8oclenc_hip_fast")]


def write_source(path: str, W: int, H: int, nframes: int, nd: int = 8, seed: int = 1) -> int:
    """configs[2]'s input as a .y4m: the synthetic sequence bench.py's legs cycle through (bench_frames), `nframes` frames"""
    from vp8oclenc_amd.synth import bench_frames
    _, _, source, _ = bench_frames(W, H, seed, nd)
    blobs = [b"FRAME\n" + b"".join(p.tobytes() for p in f) for f in source]
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F30:1 Ip A1:1 C420jpeg XYSCSS=420JPEG\n".encode())
        for t in range(nframes):
            f.write(blobs[t % nd])
    return os.path.getsize(path)


def build_y4m_to_ivf(tmp: str) -> str | None:
    if shutil.which("g++") is None:
        return None
    exe, lib = os.path.join(tmp, "y4m_to_ivf"), os.path.join(ROOT, "vp8oclenc_amd")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "native", "y4m_to_ivf.cpp"), "-o", exe,
                        "-L", lib, "-lvp8hip", "-Wl,-rpath," + lib], capture_output=True, text=True, timeout=300)
    return exe if r.returncode == 0 else None


def run_once(cmd, env, cwd, out_path, timeline=False):
    e = dict(env)
    if timeline:
        e["VP8HIP_DROP_IN_TIMELINE"] = "1"
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=e, cwd=cwd)
    wall = time.perf_counter() - t0
    if r.returncode not in (0, 777 & 255):      # the reference's main() returns 777
        return {"error": f"exit {r.returncode}: {(r.stdout[-300:] + r.stderr[-300:])!r}"}
    m = re.search(r"frame loop (\d+) frames ([0-9.]+) s", r.stderr + r.stdout)
    res = {"process_seconds": round(wall, 4), "loop_frames": int(m.group(1)) if m else None, "loop_seconds": float(m.group(2)) if m else None,
           "sha256": hashlib.sha256(open(out_path, "rb").read()).hexdigest(), "bytes": os.path.getsize(out_path)}
    if timeline:
        res["timeline"] = [l.split("vp8hip_drop_in: ", 1)[1] for l in r.stderr.splitlines() if l.startswith("vp8hip_drop_in: in ")]
    return res


def measure(W=1920, H=1080, nframes=300, repeats=3, opts=(), programs=None, timeline=True) -> dict:
    have = [(n, os.path.join(REFDIR, b)) for n, b in PROGRAMS if os.path.exists(os.path.join(REFDIR, b)) and (programs is None or n in programs)]
    if not have:
        return {"skipped": "oracle/_ref/vp8oclenc_hip* absent: oracle/ref_main/build.sh builds them where the reference checkout exists"}
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    tmp = tempfile.mkdtemp(prefix="vp8_drop_in_", dir=base)
    bindir = tempfile.mkdtemp(prefix="vp8_drop_in_bin_")        # (/dev/shm is mounted noexec on the GPU boxes: the comparator is built elsewhere)
    try:
        src = os.path.join(tmp, "in.y4m")
        size = write_source(src, W, H, nframes)
        lib = os.path.join(ROOT, "vp8oclenc_amd")
        env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
        out = {"workload": f"{W}x{H} YUV420, {nframes} frames, the reference's defaults" + (" " + " ".join(opts) if opts else "") + " (-g 150, LAST+GOLDEN+ALTREF, check_SSIM, "
                           "loop filter, one partition), .y4m -> .ivf on tmpfs, a fresh process per run",
               "source_bytes": size, "tmpfs": base is not None, "repeats": repeats, "programs": {}}
        mine = build_y4m_to_ivf(bindir)
        ref_sha = None
        if mine:
            runs = []
            for i in range(repeats):
                o = os.path.join(tmp, "mine.ivf")
                t0 = time.perf_counter()
                r = subprocess.run([mine, src, o] + list(opts), capture_output=True, text=True, timeout=900, env=env)
                wall = time.perf_counter() - t0
                if r.returncode != 0:
                    runs.append({"error": (r.stdout + r.stderr)[-300:]})
                    continue
                m = re.search(r"([0-9.]+) s of reading \+ coding \+ writing", r.stdout + r.stderr)
                runs.append({"process_seconds": round(wall, 4), "loop_seconds": float(m.group(1)) if m else None,
                             "sha256": hashlib.sha256(open(o, "rb").read()).hexdigest(), "bytes": os.path.getsize(o)})
            good = [x for x in runs if "error" not in x]
            if good:
                ref_sha = good[0]["sha256"]
                best = min(good, key=lambda x: x["process_seconds"])
                out["programs"]["y4m_to_ivf"] = {"what": "the product's own loop (vp8drv_*): the comparator and the bytes", "fps_process": round(nframes / best["process_seconds"], 1),
                                                 "fps_loop": None if not best["loop_seconds"] else round(nframes / best["loop_seconds"], 1), "runs": runs}
        for name, exe in have:
            runs = []
            for i in range(repeats):
                runs.append(run_once([exe, "-i", src, "-o", os.path.join(tmp, name + ".ivf")] + list(opts), env, tmp, os.path.join(tmp, name + ".ivf"),
                                     timeline=timeline and i == repeats - 1))
            good = [x for x in runs if "error" not in x and x["loop_seconds"]]
            entry = {"binary": os.path.relpath(exe, ROOT), "runs": runs}
            if good:
                best = min(good[:repeats - 1] or good, key=lambda x: x["loop_seconds"])      # (the last run carries the per-call clocks: not the one quoted)
                entry.update(fps_loop=round(best["loop_frames"] / best["loop_seconds"], 1), fps_process=round(best["loop_frames"] / best["process_seconds"], 1),
                             ms_per_frame_loop=round(best["loop_seconds"] / best["loop_frames"] * 1e3, 4),
                             identical_to_y4m_to_ivf=None if ref_sha is None else all(x["sha256"] == ref_sha for x in good))
                tl = [x.get("timeline") for x in runs if x.get("timeline")]
                if tl:
                    entry["host_timeline_of_the_last_run"] = tl[-1]
            out["programs"][name] = entry
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
        shutil.rmtree(bindir, ignore_errors=True)




def drop_in_leg(args) -> dict:
    """bench.py's leg: configs[2] through every program there is; never a reason to lose the bench line"""
    try:
        frames = int(os.environ.get("VP8_BENCH_DROP_IN_FRAMES", "300"))
        return measure(args.width, args.height, frames, repeats=3)
    except Exception as e:
        return {"error": repr(e)[:300]}
