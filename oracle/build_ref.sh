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
# What this is NOT: the reference host program (vp8enc.cpp) builds here but cannot run (the
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

# The reference's HOST intra path (src/intra_part.h, check_SSIM in src/vp8enc.cpp): the translation unit is
# compiled from where it lies, main() renamed, driven by oracle/ref_host_driver.cpp.  It needs the OpenCL headers
# (ROCm ships them) and libOpenCL (the ICD loader of this image) only to link; no OpenCL call is reached.
CXX=${CXX:-g++}
$CXX -O2 -fPIC -shared -ffp-contract=off -w -DCL_TARGET_OPENCL_VERSION=120 -I"$REF/src" -I/opt/rocm/include \
    "$HERE/ref_host_driver.cpp" "$REF/src/entropy_host.cpp" -o "$OUT/libvp8refhost.so" -lOpenCL
echo "built $OUT/libvp8refhost.so"
