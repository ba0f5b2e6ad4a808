#!/bin/sh
# build.sh -- the reference's OWN host program (src/vp8enc.cpp: main(), ParseArgs, the YUV4MPEG2 reader, the frame-type state
# machine, prepare_segments_data, scene_change, the IVF writer; src/entropy_host.cpp) built against libvp8hip.so.
#
# TEST INFRASTRUCTURE ONLY; runs only where the reference checkout exists (this container).  It applies the patch of
# INTEGRATION.md section 2 to a temporary copy of $REF/src -- line-addressed sed edits that delete the OpenCL traffic and call
# the functions of vp8hip_drop_in.h (this directory) in its place -- and compiles the result with g++ into
#     oracle/_ref/vp8oclenc_hip        every stage on the device (INTEGRATION.md's optional blocks; no OpenCL device needed)
#     oracle/_ref/vp8oclenc_hip_host   -DVP8HIP_KEEP_HOST_STAGES: the reference's host intra path, check_SSIM and encode_header stay
#     oracle/_ref/vp8oclenc_hip_fast   -DVP8HIP_FAST: the device build with the library's asynchronous entry points and device-side scans where
#                                      the reference's loop blocks or scans on the host; a frame's bytes leave one iteration late
# Nothing of the reference is stored in this repo: the copy lives in a mktemp directory that is removed on exit, and only the
# three binaries land in oracle/_ref/ (git-ignored; they travel to the GPU box like the other _ref artefacts).
# The binaries take the reference's own command line (-i in.y4m -o out.ivf -g .. -partitions .. -SSIM-target .. ...).
#
# The edits, by reference line (each sed expression below names what it removes):
#   vp8enc.cpp   12a      include the drop-in header behind the globals
#                14-35    ifFlush / finalFlush (clFlush wrappers, no caller left)
#                50-91    entropy_encode(): count_probs ... encode_coefficients on the CPU device, encode_header
#                222-227  prepare_segments_data(): the -g 1 early return and clEnqueueWriteBuffer(segments_data_gpu / _cpu)
#                131-227  prepare_segments_data(): all of it, with get_loopfilter_strength's scan      (fast build: on the device)
#                270-284  scene_change(): the two sums over the chroma planes                         (fast build: on the device)
#                486,489  main(): write_output_file() one iteration late, and once more behind the loop (fast build)
#                242-257  check_SSIM(): the per-macroblock fallback loop and the mean            (device build only)
#                353-362  main(): clFinish + clEnqueueMapBuffer of the coefficient and reconstruction buffers
#                386-406  main(): upload of the current frame (and, LF on the CPU device, of the filtered reconstruction)
#                421-434  main(): the read-backs after inter_transform
#                439-440  main(): clFinish x2
#                457-470  main(): unmap + upload of coefficients / parts / segment ids to the filtering device
#                505-681  finalize(): clRelease*
#   init.h       23-100   cl_info(); 1311 its call
#                107-374  init_all(): platforms, devices, contexts, programs, kernels
#                430-1276 init_all(): buffers, images, kernel arguments, queues
#   inter_part.h 1-384    prepare_GPU_buffers() + inter_transform(): all of it
#   loop_filter.h 1-190   all of it
#   intra_part.h 1100-1126 intra_transform(): the host loop and the uploads (device build) / 1112-1126 the uploads (host build)
#   encIO.h      3-28     gather_frame()'s body (device build) / 4, 24-25 its read-backs -> memcpy (host build)
#                206-253  get_yuv420_frame()'s body: fread, copy_with_padding, the chroma copies for scene_change (fast build: a reader thread)
#   debug.h      12-23    dump(): read-back of the filtered reconstruction
set -e
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
ROOT=$(cd "$HERE/../.." && pwd)
OUT="$ROOT/oracle/_ref"
CXX=${CXX:-g++}
[ -f "$REF/src/vp8enc.cpp" ] || { echo "ref_main/build.sh: no reference at $REF" >&2; exit 1; }
[ -f "$ROOT/vp8oclenc_amd/libvp8hip.so" ] || { echo "ref_main/build.sh: build libvp8hip.so first (python -m vp8oclenc_amd.build)" >&2; exit 1; }
mkdir -p "$OUT"

build() {   # $1 = device | host | fast, $2 = output name, $3 = extra compiler flags
    TMP=$(mktemp -d)
    trap 'rm -rf "$TMP"' EXIT
    cp "$REF/src/vp8enc.h" "$REF/src/entropy_host.h" "$REF/src/entropy_host.cpp" "$TMP/"
    # ---- vp8enc.cpp
    cat > "$TMP/vp8enc.sed" <<'SED'
12a #include "vp8hip_drop_in.h"
14,35d
50,91c\
	hip_entropy_encode();
353,362d
386,406c\
			hip_upload_current();
421,434c\
				hip_download_results();
439,440d
457,470c\
		hip_upload_host_results();
505,681c\
	hip_finalize();
SED
    if [ "$1" = host ]; then
        printf '1112,1126c\\\n\thip_upload_intra_results();\n' > "$TMP/intra.sed"
        cat > "$TMP/encio.sed" <<'SED'
4d
24,25c\
		memcpy(&frames.encoded_frame[frames.encoded_frame_size], frames.partitions + i*video.partition_step, frames.partition_sizes[i]);
SED
    else
        cat >> "$TMP/vp8enc.sed" <<'SED'
242,257c\
	hip_check_ssim(&min1, &min2);
SED
        printf '1100,1126c\\\n\thip_intra_transform();\n' > "$TMP/intra.sed"
        printf '3,28d\n' > "$TMP/encio.sed"
    fi
    if [ "$1" = fast ]; then
        cat >> "$TMP/vp8enc.sed" <<'SED'
131,227c\
	hip_prepare_segments(update_filter, shrpnss);
270,284c\
	int Udiff = 0, Vdiff = 0; hip_chroma_diffs(&Udiff, &Vdiff);
486c\
		if (hip_have_lagged) { --frames.frame_number; write_output_file(); ++frames.frame_number; }
489i\
	hip_flush_last_frame(); if (hip_have_lagged) { --frames.frame_number; write_output_file(); ++frames.frame_number; }
SED
        cat >> "$TMP/encio.sed" <<'SED'
206,253c\
	return hip_get_frame();
SED
    else
        cat >> "$TMP/vp8enc.sed" <<'SED'
222,227c\
	hip_set_segments();
SED
    fi
    sed -f "$TMP/vp8enc.sed" "$REF/src/vp8enc.cpp" > "$TMP/vp8enc.cpp"
    # ---- init.h
    cat > "$TMP/init.sed" <<'SED'
23,100d
1311d
107,374d
430,1276c\
	if (hip_init() < 0) return -1;
SED
    sed -f "$TMP/init.sed" "$REF/src/init.h" > "$TMP/init.h"
    : > "$TMP/inter_part.h"
    : > "$TMP/loop_filter.h"
    sed -f "$TMP/intra.sed" "$REF/src/intra_part.h" > "$TMP/intra_part.h"
    sed -f "$TMP/encio.sed" "$REF/src/encIO.h" > "$TMP/encIO.h"
    printf '12,23c\\\n\thip_download_last();\n' > "$TMP/debug.sed"
    sed -f "$TMP/debug.sed" "$REF/src/debug.h" > "$TMP/debug.h"
    # the OpenCL headers stay (the reference's structs are made of cl_int / cl_mem members); no OpenCL library is linked
    $CXX -O2 -w -fpermissive -DCL_TARGET_OPENCL_VERSION=120 $3 -I"$TMP" -I"$HERE" -I"$ROOT/include" -I/opt/rocm/include \
        "$TMP/vp8enc.cpp" "$TMP/entropy_host.cpp" -L"$ROOT/vp8oclenc_amd" -lvp8hip -Wl,-rpath,'$ORIGIN/../../vp8oclenc_amd' -o "$OUT/$2"
    [ -n "$KEEP_PATCHED" ] && cp -r "$TMP" "$KEEP_PATCHED.$1"     # (for looking at the patched files while working on the edits; never committed)
    rm -rf "$TMP"
    trap - EXIT
    echo "built $OUT/$2"
}
build device vp8oclenc_hip ""
build host vp8oclenc_hip_host "-DVP8HIP_KEEP_HOST_STAGES"
build fast vp8oclenc_hip_fast "-DVP8HIP_FAST -pthread"
