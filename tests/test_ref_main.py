"""The drop-in proved with the reference's OWN main(): oracle/ref_main/build.sh applies the patch of INTEGRATION.md section 2 to
a temporary copy of /root/reference/src (line-addressed sed edits; nothing of the reference is stored here) and compiles it
against libvp8hip.so -- once with every stage on the device, once with the reference's host intra path, check_SSIM and
encode_header kept (oracle/_ref/vp8oclenc_hip, vp8oclenc_hip_host).  CPU: they build and link, and ask for no OpenCL library.
GPU: the reference's program, driving the library through the reference's own frame loop, command line, YUV4MPEG2 reader, scene
detection and IVF writer, writes byte for byte the file scripts/native/y4m_to_ivf.cpp (the product's own loop) writes."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BIN = os.path.join(ROOT, "oracle", "_ref", "vp8oclenc_hip")
BIN_HOST = os.path.join(ROOT, "oracle", "_ref", "vp8oclenc_hip_host")
BIN_FAST = os.path.join(ROOT, "oracle", "_ref", "vp8oclenc_hip_fast")     # -DVP8HIP_FAST: asynchronous entry points, device-side scans, a reader thread


@pytest.mark.skipif(not os.path.isdir(REF), reason="no reference checkout here (the GPU box uses the prebuilt binaries)")
def test_the_references_main_builds_against_the_library():
    from vp8oclenc_amd import api
    api.load_library()      # (builds libvp8hip.so if stale)
    r = subprocess.run(["sh", os.path.join(ROOT, "oracle", "ref_main", "build.sh"), REF], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    for exe in (BIN, BIN_HOST, BIN_FAST):
        assert os.path.exists(exe)
        dyn = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
        assert "libvp8hip.so" in dyn and "OpenCL" not in dyn, dyn       # the library instead of the OpenCL loader, not beside it
        und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
        calls = sorted({l.split()[-1] for l in und.splitlines() if " vp8hip_" in l or " vp8host_" in l})
        assert not [l for l in und.splitlines() if " cl" in l and l.split()[-1].startswith("cl") and not l.split()[-1].startswith("clock_")], und    # no OpenCL entry point is referenced
        assert {"vp8hip_create", "vp8hip_destroy", "vp8hip_upload_current", "vp8hip_inter_transform", "vp8hip_prepare_filter_mask", "vp8hip_loop_filter"} <= set(calls), calls
        assert ("vp8hip_set_segments" in calls) == (exe != BIN_FAST), calls
    fast = subprocess.run(["nm", "-D", "--undefined-only", BIN_FAST], capture_output=True, text=True).stdout
    for sym in ("vp8hip_prefetch_current", "vp8hip_auto_segments", "vp8hip_chroma_change", "vp8hip_check_ssim_async", "vp8hip_check_ssim_result",
                "vp8hip_encode_frame_begin", "vp8hip_encode_frame_end", "vp8hip_set_source_size", "vp8hip_host_alloc"):
        assert sym in fast, sym
    assert " vp8hip_check_ssim\n" not in fast and " vp8hip_encode_frame\n" not in fast      # the blocking forms are gone from the fast build
    host_only = subprocess.run(["nm", "-D", "--undefined-only", BIN_HOST], capture_output=True, text=True).stdout
    assert "vp8hip_download_results" in host_only and "vp8hip_upload_mb_data" in host_only and "vp8hip_encode_coefficients" in host_only


def _y4m(path, W, H, frames, cut=None, seed=3):
    from vp8oclenc_amd import y4m
    from vp8oclenc_amd.synth import SynthSequence
    a, b = SynthSequence(W, (H + 15) // 16 * 16, seed=seed), SynthSequence(W, (H + 15) // 16 * 16, seed=seed + 87)

    def crop(planes, du=0):
        y, u, v = planes
        return (np.ascontiguousarray(y[:H, :W]), np.ascontiguousarray(np.clip(u[:H // 2, :W // 2].astype(int) + du, 0, 255).astype(np.uint8)),
                np.ascontiguousarray(v[:H // 2, :W // 2]))
    src = [crop(a.frame(t)) if cut is None or t < cut else crop(b.frame(t - cut), 45) for t in range(frames)]
    y4m.write_y4m(path, src, framerate=25)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,frames,cut,opts", [
    (320, 192, 9, None, []),                                                              # the defaults: -g 150, one partition, no SSIM target
    (320, 184, 12, 6, ["-g", "9", "-partitions", "2"]),                                   # eight rows of padding, a scene cut, a key frame at the GOP boundary
    (320, 192, 10, None, ["-g", "6", "-partitions", "4", "-qmin", "40", "-qmax", "110", "-SSIM-target", "92", "-altref-range", "3"]),   # replaced macroblocks
    (176, 144, 8, 4, ["-qmin", "50", "-qmax", "110", "-SSIM-target", "90"]),              # a frame sent back to be a key frame
    (1920, 1080, 4, None, ["-partitions", "8"]),
    (1920, 1080, 40, 23, ["-g", "16"]),                                                   # the metric's geometry: key frames at GOP boundaries, a cut, golden and altref periods
    (352, 288, 30, None, ["-g", "1"]),                                                    # every frame a key frame
    (1920, 1080, 14, None, ["-g", "7", "-qmin", "40", "-qmax", "110", "-SSIM-target", "93", "-partitions", "2"]),   # the metric's geometry with the four-pass ladder and replaced macroblocks
])
def test_the_references_main_writes_the_products_file(tmp_path, W, H, frames, cut, opts):
    if not (os.path.exists(BIN) and os.path.exists(BIN_HOST)):
        pytest.skip("oracle/_ref/vp8oclenc_hip not built (oracle/ref_main/build.sh needs the reference checkout)")
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "y4m_to_ivf")
    lib = os.path.join(ROOT, "vp8oclenc_amd")
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "native", "y4m_to_ivf.cpp"), "-o", exe,
                    "-L", lib, "-lvp8hip", "-Wl,-rpath," + lib], check=True, timeout=300)
    src = str(tmp_path / "in.y4m")
    _y4m(src, W, H, frames, cut)
    env = dict(os.environ, LD_LIBRARY_PATH=lib + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    outs = {}
    for name, cmd in (("device", [BIN, "-i", src, "-o", str(tmp_path / "ref_device.ivf")] + opts),
                      ("host", [BIN_HOST, "-i", src, "-o", str(tmp_path / "ref_host.ivf")] + opts),
                      ("fast", [BIN_FAST, "-i", src, "-o", str(tmp_path / "ref_fast.ivf")] + opts)):
        if name == "fast" and not os.path.exists(BIN_FAST):
            continue
        if name == "host" and W > 1000:
            continue          # (the reference's host intra path takes a tenth of a second per 1080p key frame: not the point here)
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
        assert r.returncode in (0, 777 & 255), (name, r.returncode, r.stdout[-1500:], r.stderr[-1500:])     # main() returns 777
        outs[name] = open(cmd[4], "rb").read()
        assert "scene changes detected" in r.stdout
        if cut is not None and "-SSIM-target" not in opts:
            assert "1 scene changes detected by color change" in r.stdout, r.stdout[-600:]
    mine = opts[:]
    if "-SSIM-target" in mine:       # the reference takes the target in hundredths (init.h:1512)
        i = mine.index("-SSIM-target")
        mine[i + 1] = str(int(mine[i + 1]) / 100.0)
    r = subprocess.run([exe, src, str(tmp_path / "mine.ivf")] + mine, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    mine_bytes = open(tmp_path / "mine.ivf", "rb").read()
    assert len(mine_bytes) > 32 + 12 * frames
    for name, data in outs.items():
        assert data[:32] == mine_bytes[:32], f"{name}: IVF header"
        assert data == mine_bytes, f"the reference's main() ({name} build) and y4m_to_ivf differ: {len(data)} vs {len(mine_bytes)} bytes, first at " \
                                   f"{next((i for i, (p, q) in enumerate(zip(data, mine_bytes)) if p != q), -1)}"
    if "-SSIM-target" in opts and W == 176:
        assert "recoded" in r.stdout and " 0 recoded" not in r.stdout, r.stdout
