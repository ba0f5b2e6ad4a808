"""File in, file out, and back: scripts/encode_ivf.py (raw I420 or the synthetic sequence -> GOP chunks through the native
frame loop -> .ivf) and scripts/decode_ivf.py (the tests' RFC 6386 decoder).  What comes back must LOOK like what went in
(luma PSNR per frame), which none of the byte-level pins says, and with --conformant must be the encoder's reconstruction."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _run(*args):
    r = subprocess.run([sys.executable, *args], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("W,H,frames,gop,extra,floor", [
    (320, 192, 12, 5, [], 31.0),
    (640, 352, 8, 4, ["--partitions", "4", "--qmin", "40", "--qmax", "100"], 29.0),
    (320, 192, 10, 5, ["--conformant", "--ssim-target", "0.92", "--qmin", "40", "--qmax", "110", "--partitions", "2"], 29.0),
    (500, 300, 7, 4, ["--partitions", "2"], 31.0),          # not multiples of 16: padded on the device, display size in the key frames
])
def test_encode_ivf_then_decode_gives_the_pictures_back(tmp_path, W, H, frames, gop, extra, floor):
    import decode_ivf
    import vp8_decode
    from vp8oclenc_amd.synth import SynthSequence
    seq = SynthSequence(W, H, seed=1)
    if W % 16 or H % 16:      # the synthetic source is made at multiples of 16: cut the picture out of a larger one
        big = SynthSequence(W + 16, H + 16, seed=1)

        class Crop:
            def frame(self, t):
                y, u, v = big.frame(t)
                return (np.ascontiguousarray(y[:H, :W]), np.ascontiguousarray(u[:H // 2, :W // 2]), np.ascontiguousarray(v[:H // 2, :W // 2]))
        seq = Crop()
        seq.W, seq.H = W, H
    yuv = tmp_path / "in.yuv"
    with open(yuv, "wb") as f:                       # the raw I420 file the reference reads (encIO.h:141-196)
        for t in range(frames):
            for p in seq.frame(t):
                f.write(np.ascontiguousarray(p).tobytes())
    ivf = tmp_path / "out.ivf"
    out = _run(os.path.join(ROOT, "scripts", "encode_ivf.py"), str(ivf), "--yuv", str(yuv), "--width", str(seq.W), "--height", str(seq.H),
               "--frames", str(frames), "--gop", str(gop), *extra)
    assert f"{frames} frames" in out
    Wf, Hf, rate, scale, packets = decode_ivf.read_ivf(str(ivf))
    assert (Wf, Hf, len(packets)) == (seq.W, seq.H, frames) and rate == 30
    dec = vp8_decode.Decoder()
    keys = 0
    for t, fr in enumerate(packets):
        f, (Y, U, V) = dec.decode(fr)
        keys += int(f.key)
        assert f.key == (t % gop == 0) or "--ssim-target" in extra, t          # GOP chunks start with their key frame
        y, u, v = seq.frame(t)
        Y, U, V = Y[:seq.H, :seq.W], U[:seq.H // 2, :seq.W // 2], V[:seq.H // 2, :seq.W // 2]      # the display size
        assert (f.width, f.height) == (seq.W, seq.H) or not f.key
        p = decode_ivf.psnr(Y, y)
        assert p > floor and decode_ivf.psnr(U, u) > floor and decode_ivf.psnr(V, v) > floor, (t, p)
    assert keys >= (frames + gop - 1) // gop
    # the same file through the command-line decoder (its PSNR report against the synthetic source)
    rep = _run(os.path.join(ROOT, "scripts", "decode_ivf.py"), str(ivf), "--yuv", str(yuv), "--out", str(tmp_path / "dec.yuv"))
    assert "lowest luma PSNR" in rep and os.path.getsize(tmp_path / "dec.yuv") == frames * seq.W * seq.H * 3 // 2


def test_y4m_in_ivf_out(tmp_path):
    """the reference's own input format: size and frame rate from the YUV4MPEG2 header (vp8oclenc_amd/y4m.py), 354x290 frames
    (padded to 368x304 on the device), frame rate 25 into the IVF header"""
    import decode_ivf
    import vp8_decode
    from vp8oclenc_amd import y4m
    from vp8oclenc_amd.synth import SynthSequence
    W, H, frames = 354, 290, 6
    big = SynthSequence(W + 16, H + 16, seed=2)
    src = [tuple(np.ascontiguousarray(p[:h, :w]) for p, (h, w) in zip(big.frame(t), ((H, W), (H // 2, W // 2), (H // 2, W // 2)))) for t in range(frames)]
    y4m.write_y4m(str(tmp_path / "in.y4m"), src, framerate=25)
    ivf = tmp_path / "out.ivf"
    _run(os.path.join(ROOT, "scripts", "encode_ivf.py"), str(ivf), "--y4m", str(tmp_path / "in.y4m"), "--gop", "4", "--partitions", "2")
    Wf, Hf, rate, scale, packets = decode_ivf.read_ivf(str(ivf))
    assert (Wf, Hf, rate, len(packets)) == (W, H, 25, frames)
    dec = vp8_decode.Decoder()
    for t, fr in enumerate(packets):
        f, (Y, U, V) = dec.decode(fr)
        assert (f.mbw, f.mbh) == (23, 19) and ((f.width, f.height) == (W, H) or not f.key)
        assert decode_ivf.psnr(Y[:H, :W], src[t][0]) > 31.0, t


def test_native_y4m_to_ivf_program(tmp_path):
    """scripts/native/y4m_to_ivf.cpp: the reference's program with the path swapped in, as plain C++ over the C ABI (built
    here with g++: the headers are C).  Its .ivf == the frames of the same loop driven through ctypes, byte for byte, with a
    scene cut found by -scene-detect and the source padded on the device."""
    import shutil
    import decode_ivf
    from vp8oclenc_amd import api, y4m
    from vp8oclenc_amd.synth import SynthSequence
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "y4m_to_ivf")
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "native", "y4m_to_ivf.cpp"), "-o", exe,
                    "-L", os.path.join(ROOT, "vp8oclenc_amd"), "-lvp8hip", "-Wl,-rpath," + os.path.join(ROOT, "vp8oclenc_amd")], check=True, timeout=300)
    W, H = 360, 200
    a, b = SynthSequence(W + 16, H + 16, seed=3), SynthSequence(W + 16, H + 16, seed=90)

    def crop(planes, du=0):
        y, u, v = planes
        return (np.ascontiguousarray(y[:H, :W]), np.ascontiguousarray(np.clip(u[:H // 2, :W // 2].astype(int) + du, 0, 255).astype(np.uint8)),
                np.ascontiguousarray(v[:H // 2, :W // 2]))
    src = [crop(a.frame(t)) for t in range(6)] + [crop(b.frame(t), 45) for t in range(6)]      # a cut at frame 6
    y4m.write_y4m(str(tmp_path / "in.y4m"), src, framerate=24)
    r = subprocess.run([exe, str(tmp_path / "in.y4m"), str(tmp_path / "out.ivf"), "-g", "9", "-partitions", "2", "-scene-detect"], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"})      # a C++ host that knows nothing of hardware queues
    assert r.returncode == 0, r.stdout + r.stderr
    assert "16 hardware queues" in r.stdout, r.stdout       # the library set them when it was loaded (csrc/api_context.hip)
    assert "12 frames 360x200 (coded 368x208)" in r.stdout and "(1 by scene change" in r.stdout, r.stdout
    Wf, Hf, rate, scale, packets = decode_ivf.read_ivf(str(tmp_path / "out.ivf"))
    assert (Wf, Hf, rate, len(packets)) == (W, H, 24, 12)
    assert int.from_bytes(open(tmp_path / "out.ivf", "rb").read(32)[24:28], "little") == 13          # the frame count as the reference writes it: one too many
    drv = api.NativeDriver(368, 208, gop_size=9, num_partitions=2, scene_detect=1, src_width=W, src_height=H)
    for t, f in enumerate(src):
        drv.encode_frame_host(*f)
        assert drv.get_frame() == packets[t], t
    assert drv.stats().scene_changes == 1
    drv.close()


@pytest.mark.parametrize("W,H,frames,gop,chunks,batch,partitions", [(360, 200, 50, 7, 5, 3, 2), (320, 192, 23, 5, 48, 6, 1), (640, 352, 36, 12, 2, 1, 4)])
def test_native_gop_parallel_program_writes_the_serial_programs_file(tmp_path, W, H, frames, gop, chunks, batch, partitions):
    """scripts/native/y4m_to_ivf_gops.cpp: one Y4M file with its closed GOPs coded SIDE BY SIDE -- `chunks` of them in flight in batches of
    `batch`, a host thread per batch, the frames from page-locked host memory with the next frame's copy under way while the current one
    is coded, drivers reused round after round, members sitting out where the file ends -- writes, byte for byte, the file the one-video
    program (scripts/native/y4m_to_ivf.cpp: the reference's loop, frame after frame) writes with the same -g.  SURVEY 8(e): a key frame
    resets every reference (intra_part.h:1091-1098), the host concatenates the frames in order."""
    import shutil
    from vp8oclenc_amd import y4m
    from vp8oclenc_amd.synth import SynthSequence
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exes = {}
    for name in ("y4m_to_ivf", "y4m_to_ivf_gops"):
        exes[name] = str(tmp_path / name)
        subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "native", name + ".cpp"), "-o", exes[name],
                        "-L", os.path.join(ROOT, "vp8oclenc_amd"), "-lvp8hip", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "vp8oclenc_amd")], check=True, timeout=300)
    seq = SynthSequence(W + 16, H + 16, seed=17)
    src = []
    for t in range(frames):
        y, u, v = seq.frame(t)
        src.append((np.ascontiguousarray(y[:H, :W]), np.ascontiguousarray(u[:H // 2, :W // 2]), np.ascontiguousarray(v[:H // 2, :W // 2])))
    y4m.write_y4m(str(tmp_path / "in.y4m"), src, framerate=25)
    common = ["-g", str(gop), "-partitions", str(partitions)]
    a = subprocess.run([exes["y4m_to_ivf"], str(tmp_path / "in.y4m"), str(tmp_path / "serial.ivf"), "-no-scene-detect"] + common, capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stdout + a.stderr
    b = subprocess.run([exes["y4m_to_ivf_gops"], str(tmp_path / "in.y4m"), str(tmp_path / "gops.ivf"), "-chunks", str(chunks), "-batch", str(batch)] + common,
                       capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stdout + b.stderr
    nchunks = (frames + gop - 1) // gop
    assert f"{frames} frames {W}x{H}" in b.stdout and f"in {nchunks} closed GOPs of {gop}" in b.stdout and f"{nchunks} key frames" in b.stdout, b.stdout
    one, many = open(tmp_path / "serial.ivf", "rb").read(), open(tmp_path / "gops.ivf", "rb").read()
    assert len(one) == len(many) and one == many, f"{len(one)} vs {len(many)} bytes, first difference at {next((i for i, (p, q) in enumerate(zip(one, many)) if p != q), None)}"


def test_native_gop_parallel_program_with_frames_sent_back_still_writes_a_stream_that_decodes(tmp_path):
    """y4m_to_ivf_gops with an SSIM target: check_SSIM replaces macroblocks and sends frames back to be key frames INSIDE chunks (the
    serial run's key frames would move then, so the file is not compared with the serial program's) -- every chunk still starts with its
    key frame, drivers are reused round after round, and the conformant stream decodes (tests' RFC 6386 decoder) to pictures that look
    like the source, frame by frame."""
    import shutil
    import decode_ivf
    import vp8_decode
    from vp8oclenc_amd import y4m
    from vp8oclenc_amd.synth import SynthSequence
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "y4m_to_ivf_gops")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "scripts", "native", "y4m_to_ivf_gops.cpp"), "-o", exe,
                    "-L", os.path.join(ROOT, "vp8oclenc_amd"), "-lvp8hip", "-lpthread", "-Wl,-rpath," + os.path.join(ROOT, "vp8oclenc_amd")], check=True, timeout=300)
    W, H, frames, gop = 320, 192, 40, 8
    seq = SynthSequence(W, H, seed=23)
    src = [seq.frame(t) for t in range(frames)]
    y4m.write_y4m(str(tmp_path / "in.y4m"), src, framerate=30)
    r = subprocess.run([exe, str(tmp_path / "in.y4m"), str(tmp_path / "out.ivf"), "-g", str(gop), "-partitions", "2", "-chunks", "3", "-batch", "3", "-conformant",
                        "-SSIM-target", "0.93", "-qmin", "40", "-qmax", "110"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    Wf, Hf, rate, scale, packets = decode_ivf.read_ivf(str(tmp_path / "out.ivf"))
    assert (Wf, Hf, len(packets)) == (W, H, frames)
    dec = vp8_decode.Decoder()
    keys = 0
    for t, fr in enumerate(packets):
        f, (Y, U, V) = dec.decode(fr)
        keys += int(f.key)
        if t % gop == 0:
            assert f.key, t                       # every chunk starts with its key frame
        assert decode_ivf.psnr(Y[:H, :W], src[t][0]) > 24.0, t
    assert keys >= frames // gop
    assert f"{keys} key frames" in r.stdout, (keys, r.stdout)
