"""Build libvp8hip.so (HIP kernels + C ABI) for gfx950, in-tree, with hipcc."""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libvp8hip.so")
SOURCES = ["api_context.hip", "api_inter.hip", "api_entropy.hip", "api_batch.hip", "api_shard.hip", "api_profile.hip", "kernels_me.hip", "kernels_s2.hip", "kernels_mb.hip", "kernels_lf3.hip", "kernels_lf4.hip", "kernels_entropy_stage.hip", "kernels_rc.hip", "kernels_intra.hip", "vp8_host.cpp", "vp8_driver.cpp", "vp8_bitstream.cpp"]
HEADERS = ["vp8hip_ctx.h", "vp8hip_dev.h", "kernels_rc_dev.h", "vp8_rfc6386_tables.inc", "vp8_mbhdr.h", "kernels_ent.hip", "kernels_hdr.hip"]   # the two .hip files are included by kernels_entropy_stage.hip
HEADERS = HEADERS + [os.path.join("..", "..", "include", h) for h in ("vp8hip.h", "vp8hip_multi.h", "vp8hip_taps.h", "vp8hip_host.h", "vp8hip_driver.h", "vp8hip_bitstream.h")]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 ... -> vp8oclenc_amd/libvp8hip.so.  Cross-compiles without a GPU."""
    if not force and not stale():
        return LIB
    objs = []
    procs = []
    odir = os.path.join(PKG, "build")
    os.makedirs(odir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical", "-Wno-unused-value", "-Wno-missing-braces",
             "-I", os.path.join(PKG, "..", "include")] + os.environ.get("VP8HIP_EXTRA_FLAGS", "").split()
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        if not os.path.exists(path):
            continue
        obj = os.path.join(odir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        cmd = [_hipcc()] + flags + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    # (RCCL -- vp8hip_shard_*, vp8hip_group_* -- is not linked: api_shard.hip resolves it with dlopen when a host first asks for it)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
