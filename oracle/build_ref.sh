#!/bin/sh
# build_ref.sh -- compile the reference's OWN kernels for x86 into oracle/_ref/libvp8ref.so.
#
# TEST INFRASTRUCTURE ONLY; runs only where the reference checkout exists (this container).
# The sources are read from where they lie under $REF; nothing from them is stored in this
# repo: the two .cl files are piped through sed into a mktemp directory that is removed on exit,
# and only the shared object lands in oracle/_ref/ (git-ignored).
#
# Why sed: clang 22's OpenCL C front end rejects "int scalar (op) short vector", which the 2013
# AMD compiler accepted by narrowing the scalar.  The edits only add the explicit (short)/(ushort)
# casts that narrowing implied -- 4 lines of luma_search_1step (src/GPU_kernels.cl:495,499,500,556)
# and integer literals inside the loop-filter section (src/CPU_kernels.cl:829-1439).
#
# Second build (below, "gfx950"): the same two sed outputs compiled by AMD's OpenCL C compiler FOR THE MI355X, with the
# vendor's own built-in library, into code objects oracle/_ref/ref_{gpu,cpu}_kernels_gfx950.co, plus the OpenCL host
# oracle/ref_cl_driver.c -> oracle/_ref/libvp8ref_cl.so.  That pair runs the reference's kernels on the GPU box with the
# vendor's built-ins (one exception forced by the hardware: MI355X has no image unit, see ref_image_as_buffer.cl) (scripts/gen_golden_gfx950.py makes tests/golden/gfx950/*.npz from it; tests/test_gpu_refcl.py
# compares live).  Build options = the reference's own clBuildProgram options (src/init.h:181-186,337-342).
#
# What the x86 build is NOT: the reference host program (vp8enc.cpp) builds here but cannot run (the
# OpenCL platform has 0 devices), so the kernels are driven by oracle/ref_driver.c, and the
# OpenCL C built-ins they import come from oracle/ref_shim.cl (spec semantics, no encoder logic).
set -e
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/_ref"
CL=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
[ -f "$REF/src/GPU_kernels.cl" ] || { echo "build_ref: no reference at $REF" >&2; exit 1; }
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"

sed -e '495s|p = c/2;|p = c/(short)2;|' \
    -e '499s|vector /= pixel_rate;|vector /= (short)pixel_rate;|' \
    -e '500s|? 0 : vector|? (short)0 : vector|' \
    -e '556s|(vector-c)\*pixel_rate;|(vector-c)*(short)pixel_rate;|' \
    "$REF/src/GPU_kernels.cl" > "$TMP/gpu.cl"

sed -e '829,1439{' \
    -e 's/,-128,/,(short)(-128),/g' -e 's/<-128/<(short)(-128)/g' \
    -e 's/,127,/,(short)127,/g' -e 's/>127/>(short)127/g' \
    -e 's/- 128/- (short)128/g' -e 's/+ 128/+ (short)128/g' \
    -e 's/= a + 3;/= a + (short)3;/' -e 's/= a + 4;/= a + (short)4;/' \
    -e 's/(a + 1) >> 1/(a + (short)1) >> (short)1/' \
    -e 's/) \* 2 + abs/) * (ushort)2 + abs/' -e 's|) / 2)  >|) / (ushort)2)  >|' \
    -e '}' "$REF/src/CPU_kernels.cl" > "$TMP/cpu.cl"

# x86-64 baseline (no FMA): a*b+c in the kernels stays an unfused multiply and add
CLFLAGS="-x cl -cl-std=CL1.2 -Xclang -finclude-default-header -target x86_64-unknown-linux-gnu -O2 -w -fPIC -Dinline="
$CL $CLFLAGS -c "$TMP/gpu.cl" -o "$TMP/gpu.o"
$CL $CLFLAGS -DLOOP_FILTER -c "$TMP/cpu.cl" -o "$TMP/cpu.o"
CFLAGS="-target x86_64-unknown-linux-gnu -O2 -fPIC -ffp-contract=off -w"
# the built-ins: OpenCL C too (same vector calling convention), but without the default header
$CL -x cl -cl-std=CL1.2 -cl-no-stdinc -target x86_64-unknown-linux-gnu -O2 -w -fPIC -ffp-contract=off -c "$HERE/ref_shim.cl" -o "$TMP/shim.o"
$CL $CFLAGS -c "$HERE/ref_driver.c" -o "$TMP/driver.o"
$CL -target x86_64-unknown-linux-gnu -shared -o "$OUT/libvp8ref.so" "$TMP/gpu.o" "$TMP/cpu.o" "$TMP/shim.o" "$TMP/driver.o"
echo "built $OUT/libvp8ref.so"

# gfx950: vendor compiler + vendor built-ins (links opencl.bc/ocml.bc/ockl.bc from /opt/rocm/amdgcn/bitcode), no shim.
GFXFLAGS="-x cl -cl-std=CL1.0 -target amdgcn-amd-amdhsa -mcpu=gfx950 -Xclang -finclude-default-header -O3 -w"
$CL $GFXFLAGS "$TMP/gpu.cl" -o "$OUT/ref_gpu_kernels_gfx950.co"
# the MI355X has no image hardware (CL_DEVICE_IMAGE_SUPPORT = 0): for the two kernels that sample an image2d_t, a second
# build in which the texel fetch reads a buffer (ref_image_as_buffer.cl says exactly what is replaced)
$CL $GFXFLAGS -include "$HERE/ref_image_as_buffer.cl" "$TMP/gpu.cl" -o "$OUT/ref_gpu_kernels_imgbuf_gfx950.co"
$CL $GFXFLAGS -DLOOP_FILTER "$TMP/cpu.cl" -o "$OUT/ref_cpu_kernels_gfx950.co"
${CC:-gcc} -O2 -fPIC -shared -std=gnu99 -Wall -I/opt/rocm/include "$HERE/ref_cl_driver.c" -o "$OUT/libvp8ref_cl.so" -lOpenCL -ldl
echo "built $OUT/ref_gpu_kernels_gfx950.co $OUT/ref_gpu_kernels_imgbuf_gfx950.co $OUT/ref_cpu_kernels_gfx950.co $OUT/libvp8ref_cl.so"

# The reference's HOST intra path (src/intra_part.h, check_SSIM in src/vp8enc.cpp): the translation unit is
# compiled from where it lies, main() renamed, driven by oracle/ref_host_driver.cpp.  It needs the OpenCL headers
# (ROCm ships them) and libOpenCL (the ICD loader of this image) only to link; no OpenCL call is reached.
CXX=${CXX:-g++}
$CXX -O2 -fPIC -shared -ffp-contract=off -w -DCL_TARGET_OPENCL_VERSION=120 -I"$REF/src" -I/opt/rocm/include \
    "$HERE/ref_host_driver.cpp" "$REF/src/entropy_host.cpp" -o "$OUT/libvp8refhost.so" -lOpenCL
echo "built $OUT/libvp8refhost.so"
