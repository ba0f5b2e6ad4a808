"""No-GPU checks: the C-ABI library loads and exports every symbol the headers declare; the host-side
mirror (vp8_host.cpp) reproduces the reference's parameter producers and frame sequencing; the device
entry points fail loudly -- not silently fall back -- when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vp8oclenc_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("vp8hip.h", "vp8hip_multi.h", "vp8hip_taps.h", "vp8hip_host.h", "vp8hip_driver.h", "vp8hip_bitstream.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        names += re.findall(r"\b(vp8(?:hip|host|drv|bs)_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    lib = api.load_library()
    decl = declared_symbols()
    assert len(decl) >= 25
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(api.ABI_SYMBOLS) == decl, "api.ABI_SYMBOLS out of sync with include/*.h"
    # the version the header states, the library reports and the binding expects
    header = open(os.path.join(ROOT, "include", "vp8hip.h")).read()
    stated = int(re.search(r"#define VP8HIP_ABI_VERSION (\d+)", header).group(1))
    assert lib.vp8hip_abi_version() == stated == api.ABI_VERSION


def test_binding_structs_have_the_c_layout(tmp_path):
    """DrvConfig / DrvStats / the header and bitstream structs mirrored in Python against sizeof and offsetof from a C compiler"""
    import shutil
    import subprocess
    from vp8oclenc_amd import bitstream
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "s.c"
    src.write_text("""
        #include <stdio.h>
        #include <stddef.h>
        #include "vp8hip_driver.h"
        #include "vp8hip_bitstream.h"
        #include "vp8hip_host.h"
        int main(void) {
            printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(vp8drv_config), offsetof(vp8drv_config, src_height), sizeof(vp8drv_stats),
                   offsetof(vp8drv_stats, scene_changes), sizeof(vp8hip_header_params), sizeof(vp8bs_frame), sizeof(vp8host_scene_state));
            return 0;
        }""")
    exe = tmp_path / "s"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    c = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert c[0] == C.sizeof(api.DrvConfig) and c[1] == api.DrvConfig.src_height.offset
    assert c[2] == C.sizeof(api.DrvStats) and c[3] == api.DrvStats.scene_changes.offset
    assert c[5] == C.sizeof(bitstream.Frame)
    assert c[6] == C.sizeof(api.SceneState)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.Vp8HipError):
        api.Vp8Hip(64, 64)


def test_product_library_does_not_link_the_oracle():
    """The product path must not route through oracle/: no vp8o_ / ref_ symbol is linked or referenced."""
    out = os.popen(f"nm -D {api._build.LIB}").read()
    assert "vp8o_" not in out and " ref_" not in out
    for f in os.listdir(os.path.join(ROOT, "vp8oclenc_amd", "csrc")):
        src = open(os.path.join(ROOT, "vp8oclenc_amd", "csrc", f)).read()
        assert "vp8_oracle" not in src and "oracle/" not in src.replace("// oracle/", "")


def test_quantizer_ladders_reference_defaults():
    last, alt = api.quantizer_ladders(0, 48)      # init.h:1548-1603 defaults
    assert last == [12, 24, 36, 48]
    assert alt == [3, 8, 12, 24]
    last, alt = api.quantizer_ladders(60, 20)     # swapped range is corrected, altref UQ floor = qi_min
    assert last == [30, 40, 50, 60] and alt[0] == 20


def test_segments_data_matches_formulas():
    sd = api.prepare_segments_data(False, [12, 24, 36, 48], 0, reductor=4, sharpness=0)
    assert sd.shape == (4, 11)
    assert list(sd[0, 1:6]) == [15, 0, 0, -15, -15]
    dc_q = [4, 5, 6, 7, 8, 9, 10, 10, 11, 12, 13, 14, 15, 16, 17, 17, 18, 19, 20, 20, 21, 21, 22, 22, 23, 23, 24, 25,
            25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 46, 47, 48, 49,
            50, 51, 52, 53, 54, 55, 56, 57, 58]
    for i, qi in enumerate([12, 24, 36, 48]):
        lvl = dc_q[qi + 15] // 4
        assert sd[i, 0] == qi and sd[i, 6] == lvl and sd[i, 9] == max(lvl, 1)
        assert sd[i, 7] == (lvl + 2) * 2 + sd[i, 9] and sd[i, 8] == lvl * 2 + sd[i, 9]
    # sharpness clamps the interior limit; update_filter doubles the divisor and forces sharpness 7 (vp8enc.cpp:155-159)
    sd2 = api.prepare_segments_data(False, [12, 24, 36, 48], 0, reductor=4, sharpness=3, update_filter=True, shrpnss=7)
    assert (sd2[:, 6] <= sd[:, 6]).all() and (sd2[:, 9] <= 2).all() and (sd2[:, 9] >= 1).all()
    key = api.prepare_segments_data(True, [12, 24, 36, 48], 5, reductor=3, sharpness=0)
    assert (key[:, 0] == 5).all() and list(key[0, 4:6]) == [0, 0]


def test_loopfilter_strength():
    y = np.full((32, 48), 128, np.uint8)
    assert api.loopfilter_strength(y) == (128 * 5 // 255 + 3, 0)
    rng = np.random.default_rng(0)
    y = rng.integers(0, 256, size=(32, 48)).astype(np.uint8)
    red, sharp = api.loopfilter_strength(y)
    assert red == (int(round(y.mean())) * 5 // 255) + 3 or red in (5, 6)
    assert sharp == 7


def test_loopfilter_strength_matches_python_restatement():
    """vp8enc.cpp:96-127 with its int accumulators taken modulo 2^32 (large noisy frames overflow them)."""
    def ref(y):
        h, w = y.shape
        a = y.astype(np.int64)
        avg = (int(a.sum()) + w * h // 2) // (w * h)
        red = avg * 5 // 255 + 3
        nb = (a[:-2, :-2] + a[:-2, 1:-1] + a[:-2, 2:] + a[1:-1, :-2] + a[1:-1, 2:] + a[2:, :-2] + a[2:, 1:-1] + a[2:, 2:]) // 8
        div = int(((a[1:-1, 1:-1] - nb) ** 2).sum()) & 0xffffffff
        div = div - (1 << 32) if div >= (1 << 31) else div           # int32 wrap
        n = (h - 1) * (w - 1)
        div += n // 2
        div = int(div / n)                                            # C division truncates toward zero
        sh = int(div / 8)
        return red, min(sh, 7)
    rng = np.random.default_rng(5)
    for shape in ((32, 48), (144, 176), (1088, 1920)):
        y = rng.integers(0, 256, size=shape).astype(np.uint8)        # 1080p noise: the second accumulator wraps
        assert api.loopfilter_strength(y) == ref(y), shape
        y = (np.add.outer(np.arange(shape[0]), np.arange(shape[1])) // 3 % 256).astype(np.uint8)
        assert api.loopfilter_strength(y) == ref(y), shape


def test_scene_change_decision_logic():
    """vp8enc.cpp:285-310: thresholds, no two forced key frames within 4 frames, hold-over."""
    def ref_run(diffs, key_sets_detect=True):
        hold, last, out = 0, 0, []
        for n, (u, v) in enumerate(diffs):
            detect = u > 7 or v > 7 or u + v > 10
            recent = n - last < 4
            if detect and recent:
                last, hold, r = n, 1, 0
            elif detect:
                r = 1
            elif hold and recent:
                r = 0
            elif hold:
                hold, r = 0, 1
            else:
                r = 0
            if r and key_sets_detect:
                last = n                         # intra_transform, intra_part.h:1091-1098
            out.append(r)
        return out
    rng = np.random.default_rng(2)
    for trial in range(20):
        diffs = [(int(rng.integers(0, 12)), int(rng.integers(0, 12))) if rng.random() < 0.3 else (1, 2) for _ in range(60)]
        st = api.SceneState(0, 0)
        got = []
        for n, (u, v) in enumerate(diffs):
            r = api.scene_change(st, u, v, n)
            if r:
                st.last_key_detect = n
            got.append(int(r))
        assert got == ref_run(diffs), trial
        assert sum(got) > 0 or all(u <= 7 and v <= 7 and u + v <= 10 for u, v in diffs)


def test_gop_state_machine_reference_sequence():
    """Key at 0, golden = key, altref every altref_range frames (vp8enc.cpp:364-374, intra_part.h:1091-1098)."""
    g = api.Gop(gop_size=12, altref_range=5)
    seq = []
    for t in range(26):
        s = g.next()
        if s.current_is_key:
            g.key_coded()
            seq.append(("K", 0, 0))
        else:
            ug, ua = g.inter_flags()
            seq.append(("A" if s.current_is_altref else "P", ug, ua))
        g.frame_done()
    kinds = "".join(k for k, _, _ in seq)
    assert kinds.startswith("KPPPPAPPPPAPK")       # 12-frame GOP; altref at 5 and 10 after the key
    assert seq[1][1:] == (0, 0)                    # frame after a key: golden == altref == LAST
    assert seq[2][1:] == (1, 0)                    # golden usable, altref still the key frame itself
    assert seq[6][1:] == (1, 0)                    # right after an altref frame: altref == LAST
    assert seq[7][1:] == (1, 1)


def test_skip_prob():
    nz = np.array([0, 3, 0, 9, 1, 0, 0, 0], np.int32)
    assert api.skip_prob(nz) == 3 * 256 // 8
    assert api.skip_prob(np.zeros(10, np.int32)) == 2
    assert api.skip_prob(np.ones(10, np.int32)) == 254


def test_status_strings_and_bad_arguments():
    lib = api.load_library()
    assert lib.vp8hip_status_string(0) == b"ok"
    assert b"gfx950" in lib.vp8hip_status_string(-5)
    h = C.c_void_p()
    assert lib.vp8hip_create(C.byref(h), 100, 64, -1.0, 0) == -1          # not a multiple of 16
    assert lib.vp8hip_inter_transform(None, 0, 0, 0, 0) == -1
    assert lib.vp8hip_loop_filter(None) == -1
    # the host-memory entry points refuse what they cannot use before they touch a device
    for fn in (lib.vp8hip_prefetch_current, lib.vp8hip_upload_current):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        assert fn(None, None, None, None) == -1
    lib.vp8hip_batch_upload_current.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.vp8hip_batch_upload_current(None, None, None, None, None) == -1
    lib.vp8hip_batch_prefetch_current.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.vp8hip_batch_prefetch_current(None, None, None, None) == -1
    lib.vp8drv_prefetch_frame_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.vp8drv_prefetch_frame_host(None, None, None, None) == -1
    lib.vp8hip_host_alloc.argtypes = [C.c_int, C.c_size_t, C.c_void_p]
    assert lib.vp8hip_host_alloc(0, 16, None) == -1


def test_integration_section_2_is_what_the_drop_in_script_does():
    """INTEGRATION.md section 2 is generated from oracle/ref_main/build.sh + vp8hip_drop_in.h (the patch tests/test_ref_main.py
    builds and runs): the prose cannot deviate from the script"""
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_integration_section2.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_integration_section_7_names_every_environment_variable_the_library_reads():
    """INTEGRATION.md section 7 against the sources: every VP8HIP_* the library asks getenv for is in the table, and the table names none that is not read"""
    import glob
    import re
    read = set()
    for f in glob.glob(os.path.join(ROOT, "vp8oclenc_amd", "csrc", "*")):
        read |= set(re.findall(r'(?:getenv|experiment_env)\("(VP8HIP_[A-Z0-9_]+)"\)', open(f, errors="replace").read()))
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 7. Environment the library reads"):]
    named = set(re.findall(r"`(VP8HIP_[A-Z0-9_]+)(?:=\d)?`", section))
    assert read - named == set(), f"read by the library, missing from INTEGRATION.md section 7: {sorted(read - named)}"
    assert named - read - {"VP8HIP_EXPERIMENTS"} == set(), f"named in INTEGRATION.md section 7, read nowhere: {sorted(named - read)}"


def test_the_library_sets_the_hardware_queue_count_when_it_is_loaded(tmp_path):
    """The HIP runtime reads GPU_MAX_HW_QUEUES once, at the process's first HIP call; the reference creates the queues it needs itself
    (init.h:1162-1165), and a drop-in must not depend on its host's environment for that: a constructor in libvp8hip.so sets 16
    unless the host exported a value.  Checked from C (getenv is the C environment, not os.environ's snapshot): a program that
    dlopens the library with nothing exported sees 16 afterwards, and a host's own value is left alone.  Also: the library can be
    loaded where librccl is not resolvable (it is dlopened by the first vp8hip_shard_* / vp8hip_group_* call, not linked)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "q.c"
    src.write_text("""
        #include <dlfcn.h>
        #include <stdio.h>
        #include <stdlib.h>
        int main(int argc, char **argv) {
            const char *before = getenv("GPU_MAX_HW_QUEUES");
            void *h = dlopen(argv[1], RTLD_NOW);
            if (!h) { fprintf(stderr, "%s\\n", dlerror()); return 1; }
            int (*q)(void) = (int (*)(void))dlsym(h, "vp8hip_hw_queues");
            const char *after = getenv("GPU_MAX_HW_QUEUES");
            printf("%s %s %d\\n", before ? before : "-", after ? after : "-", q());
            return 0;
        }""")
    exe = tmp_path / "q"
    subprocess.run(["gcc", str(src), "-o", str(exe), "-ldl"], check=True)
    from vp8oclenc_amd import build
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([str(exe), build.LIB], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["-", "16", "16"], out
    out = subprocess.run([str(exe), build.LIB], env=dict(env, GPU_MAX_HW_QUEUES="7"), capture_output=True, text=True, check=True).stdout.split()
    assert out == ["7", "7", "7"], out
    needed = subprocess.run(["readelf", "-d", build.LIB], capture_output=True, text=True).stdout
    assert "librccl" not in needed, "RCCL must be resolved lazily (dlopen), not linked"


def test_the_stand_in_transport_exports_what_the_library_resolves(tmp_path):
    """tests/standin_rccl/standin_rccl.cpp (the shared-memory stand-in for RCCL that lets tests/test_gpu_multirank_standin.py run several
    ranks on one GPU) builds with g++ alone and exports every `nccl*` name csrc/api_shard.hip looks up with dlsym -- read from the
    source's RCCL_SYM(...) list, so a twelfth entry point there cannot be forgotten here."""
    import re
    import shutil
    import subprocess
    if shutil.which("g++") is None or not os.path.exists("/opt/rocm/include/rccl/rccl.h"):
        pytest.skip("needs g++ and the ROCm headers")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "vp8oclenc_amd", "csrc", "api_shard.hip")).read()
    wanted = {"nccl" + m for m in re.findall(r"RCCL_SYM\((\w+)\);", src)}
    assert len(wanted) >= 11, wanted
    so = str(tmp_path / "standin_rccl.so")
    subprocess.run(["g++", "-shared", "-fPIC", "-O2", "-std=c++17", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                    os.path.join(root, "tests", "standin_rccl", "standin_rccl.cpp"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-lpthread"], check=True)
    exported = {l.split()[-1] for l in subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout.splitlines() if l.strip()}
    assert wanted <= exported, wanted - exported
