/*
 * ref_cl_driver.c -- the reference's OWN kernels run on the MI355X through the vendor's OpenCL runtime.
 * TEST INFRASTRUCTURE ONLY (never linked into the product; see the header of oracle/build_ref.sh).
 *
 * oracle/build_ref.sh compiles src/GPU_kernels.cl and src/CPU_kernels.cl from where they lie under
 * /root/reference with AMD's OpenCL C compiler for gfx950 -- the vendor's built-in library (opencl.bc /
 * ocml.bc / ockl.bc) -- into code objects under oracle/_ref/.  One limit of the hardware shows: the MI355X has no
 * image unit (CL_DEVICE_IMAGE_SUPPORT = 0), so the two kernels that sample an image2d_t come from a second build in
 * which the texel fetch -- nothing else -- reads a buffer (oracle/ref_image_as_buffer.cl explains); the other
 * fourteen kernels run exactly as the reference's source compiles.  This file is the host:
 * clCreateProgramWithBinary on those code objects, the NDRanges of src/inter_part.h:5-378 and
 * src/loop_filter.h:30-32,143-170.  It exports ref_<kernel>() with the argument lists of ref_driver.c /
 * vp8_oracle.h, so tests/oracle_lib.py::Stages drives it exactly like the x86 build and like the restatement.
 *
 * Every call creates its buffers, runs ONE kernel launch on an in-order queue, reads the outputs back and
 * releases everything: slow and simple on purpose.
 */
#define CL_TARGET_OPENCL_VERSION 120
#define CL_USE_DEPRECATED_OPENCL_1_1_APIS
#define _GNU_SOURCE
#include <CL/cl.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static cl_context g_ctx;
static cl_command_queue g_q;
static cl_device_id g_dev;
static cl_program g_prog_gpu, g_prog_gpu_img, g_prog_cpu;
static int g_image_support = -1;
static char g_devname[256];

#define CK(e)                                                                         \
    do {                                                                              \
        cl_int e_ = (e);                                                              \
        if (e_ != CL_SUCCESS) {                                                       \
            fprintf(stderr, "ref_cl_driver: OpenCL error %d at %s:%d\n", e_, __FILE__, __LINE__); \
            abort();                                                                  \
        }                                                                             \
    } while (0)

static unsigned char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    *len = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *b = (unsigned char *)malloc(*len);
    if (fread(b, 1, *len, f) != *len) { free(b); b = NULL; }
    fclose(f);
    return b;
}

static cl_program load_binary(const char *dir, const char *name) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    size_t len = 0;
    unsigned char *bin = slurp(path, &len);
    if (!bin) { fprintf(stderr, "ref_cl_driver: cannot read %s\n", path); return NULL; }
    cl_int st, err;
    const unsigned char *bins[1] = {bin};
    cl_program p = clCreateProgramWithBinary(g_ctx, 1, &g_dev, &len, bins, &st, &err);
    free(bin);
    if (err != CL_SUCCESS || st != CL_SUCCESS) { fprintf(stderr, "ref_cl_driver: clCreateProgramWithBinary(%s) %d/%d\n", name, err, st); return NULL; }
    err = clBuildProgram(p, 1, &g_dev, "", NULL, NULL);
    if (err != CL_SUCCESS) {
        char log[8192] = {0};
        clGetProgramBuildInfo(p, g_dev, CL_PROGRAM_BUILD_LOG, sizeof log - 1, log, NULL);
        fprintf(stderr, "ref_cl_driver: clBuildProgram(%s) %d\n%s\n", name, err, log);
        return NULL;
    }
    return p;
}

/* 0 = ready; 1 = no OpenCL GPU device; 2 = code objects missing or rejected */
int ref_cl_init(void) {
    if (g_prog_gpu) return 0;
    cl_platform_id plats[8];
    cl_uint np = 0;
    if (clGetPlatformIDs(8, plats, &np) != CL_SUCCESS || np == 0) return 1;
    cl_uint nd = 0;
    for (cl_uint i = 0; i < np && !nd; ++i)
        if (clGetDeviceIDs(plats[i], CL_DEVICE_TYPE_GPU, 1, &g_dev, &nd) != CL_SUCCESS) nd = 0;
    if (!nd) return 1;
    clGetDeviceInfo(g_dev, CL_DEVICE_NAME, sizeof g_devname - 1, g_devname, NULL);
    cl_bool img = CL_FALSE;
    clGetDeviceInfo(g_dev, CL_DEVICE_IMAGE_SUPPORT, sizeof img, &img, NULL);
    g_image_support = img ? 1 : 0;
    cl_int err;
    g_ctx = clCreateContext(NULL, 1, &g_dev, NULL, NULL, &err);
    if (err != CL_SUCCESS) return 1;
    g_q = clCreateCommandQueue(g_ctx, g_dev, 0, &err);
    if (err != CL_SUCCESS) return 1;
    Dl_info info;
    char dir[4096] = ".";
    if (dladdr((void *)&ref_cl_init, &info) && info.dli_fname) {
        snprintf(dir, sizeof dir, "%s", info.dli_fname);
        char *s = strrchr(dir, '/');
        if (s) *s = 0; else strcpy(dir, ".");
    }
    g_prog_gpu = load_binary(dir, "ref_gpu_kernels_gfx950.co");
    g_prog_gpu_img = load_binary(dir, "ref_gpu_kernels_imgbuf_gfx950.co");
    g_prog_cpu = load_binary(dir, "ref_cpu_kernels_gfx950.co");
    if (!g_prog_gpu || !g_prog_gpu_img || !g_prog_cpu) { g_prog_gpu = NULL; return 2; }
    return 0;
}

const char *ref_cl_device_name(void) { return g_devname; }
int ref_cl_image_support(void) { return g_image_support; }   /* CL_DEVICE_IMAGE_SUPPORT of the device: 0 on MI355X */

/* ---- small helpers ---------------------------------------------------------------------------------- */
#define PAD 4096 /* work-items of a rounded-up NDRange may read a little past the arrays the host sizes exactly */

static void need(void) {
    int r = ref_cl_init();
    if (r) { fprintf(stderr, "ref_cl_driver: not initialised (%d)\n", r); abort(); }
}

static cl_mem buf_in(const void *host, size_t bytes) {
    cl_int err;
    cl_mem m = clCreateBuffer(g_ctx, CL_MEM_READ_WRITE, bytes + PAD, NULL, &err);
    CK(err);
    unsigned char zero = 0;
    CK(clEnqueueFillBuffer(g_q, m, &zero, 1, 0, bytes + PAD, 0, NULL, NULL));
    if (host && bytes) CK(clEnqueueWriteBuffer(g_q, m, CL_TRUE, 0, bytes, host, 0, NULL, NULL));
    return m;
}

static void buf_out(cl_mem m, void *host, size_t bytes) {
    if (bytes) CK(clEnqueueReadBuffer(g_q, m, CL_TRUE, 0, bytes, host, 0, NULL, NULL));
    clReleaseMemObject(m);
}

/* src/init.h:559-578 creates CL_R / CL_UNSIGNED_INT8 read-only 2D images.  The MI355X has no image hardware
 * (CL_DEVICE_IMAGE_SUPPORT = 0; clCreateImage2D returns CL_INVALID_OPERATION), so the two kernels that sample an image
 * come from the build with oracle/ref_image_as_buffer.cl in front: the "image" is a buffer whose pixel data are preceded
 * by {width, height}; the kernel argument is a sub-buffer that starts at the pixels. */
#define IMG_HDR 4096
typedef struct { cl_mem parent, pixels; } image_buf;
static image_buf image_in(const uint8_t *host, int w, int h) {
    image_buf ib;
    const size_t px = (size_t)w * h;
    cl_int err, hdr[4] = {w, h, 0, 0};
    ib.parent = clCreateBuffer(g_ctx, CL_MEM_READ_WRITE, IMG_HDR + px + PAD, NULL, &err);
    CK(err);
    CK(clEnqueueWriteBuffer(g_q, ib.parent, CL_TRUE, IMG_HDR - 16, 16, hdr, 0, NULL, NULL));
    CK(clEnqueueWriteBuffer(g_q, ib.parent, CL_TRUE, IMG_HDR, px, host, 0, NULL, NULL));
    cl_buffer_region reg = {IMG_HDR, px + PAD};
    ib.pixels = clCreateSubBuffer(ib.parent, CL_MEM_READ_WRITE, CL_BUFFER_CREATE_TYPE_REGION, &reg, &err);
    CK(err);
    return ib;
}
static void image_release(image_buf ib) { clReleaseMemObject(ib.pixels); clReleaseMemObject(ib.parent); }

static cl_kernel kern(cl_program p, const char *name) {
    cl_int err;
    cl_kernel k = clCreateKernel(p, name, &err);
    if (err != CL_SUCCESS) { fprintf(stderr, "ref_cl_driver: clCreateKernel(%s) %d\n", name, err); abort(); }
    return k;
}

#define ARG_MEM(k, i, m) CK(clSetKernelArg(k, i, sizeof(cl_mem), &(m)))
#define ARG_INT(k, i, v) do { cl_int v_ = (cl_int)(v); CK(clSetKernelArg(k, i, sizeof(cl_int), &v_)); } while (0)
#define ARG_FLT(k, i, v) do { cl_float v_ = (cl_float)(v); CK(clSetKernelArg(k, i, sizeof(cl_float), &v_)); } while (0)

static void run(cl_kernel k, size_t global, size_t local) {
    /* a pyramid level without a single 8x8 block: the host's NDRange is empty (src/inter_part.h:110; the runtime
     * refuses it with CL_INVALID_GLOBAL_WORK_SIZE and the reference carries on) -- nothing runs */
    if (global == 0) { clReleaseKernel(k); return; }
    CK(clEnqueueNDRangeKernel(g_q, k, 1, NULL, &global, local ? &local : NULL, 0, NULL, NULL));
    CK(clFinish(g_q));
    clReleaseKernel(k);
}

static size_t round256(size_t n) { return (n % 256) ? n + 256 - (n % 256) : n; }

/* ---- src/GPU_kernels.cl ----------------------------------------------------------------------------- */
void ref_downsample_x2(const uint8_t *src, uint8_t *dst, int src_w, int src_h) {
    need();
    size_t n = (size_t)src_w * src_h;
    cl_mem s = buf_in(src, n), d = buf_in(NULL, n / 4);
    cl_kernel k = kern(g_prog_gpu, "downsample_x2");
    ARG_MEM(k, 0, s); ARG_MEM(k, 1, d); ARG_INT(k, 2, src_w); ARG_INT(k, 3, src_h);
    run(k, n / 4, 0);                                   /* src/inter_part.h:11-32, local size NULL */
    buf_out(d, dst, n / 4);
    clReleaseMemObject(s);
}

/* The kernel evaluates out-of-frame candidates before masking them (src/GPU_kernels.cl:525-549): the reference
 * plane sits in the middle of a zeroed guard buffer (a sub-buffer of it is the kernel argument). */
void ref_luma_search_1step(const uint8_t *cur, const uint8_t *ref, const int16_t *src_net, int16_t *dst_net,
                           int net_width, int width, int height, int pixel_rate) {
    need();
    const size_t plane = (size_t)width * height;
    size_t guard = 4 * plane + 65536;
    guard = (guard + 4095) / 4096 * 4096;
    cl_int err;
    cl_mem big = clCreateBuffer(g_ctx, CL_MEM_READ_WRITE, plane + 2 * guard, NULL, &err);
    CK(err);
    unsigned char zero = 0;
    CK(clEnqueueFillBuffer(g_q, big, &zero, 1, 0, plane + 2 * guard, 0, NULL, NULL));
    CK(clEnqueueWriteBuffer(g_q, big, CL_TRUE, guard, plane, ref, 0, NULL, NULL));
    cl_buffer_region reg = {guard, plane + guard};
    cl_mem refm = clCreateSubBuffer(big, CL_MEM_READ_WRITE, CL_BUFFER_CREATE_TYPE_REGION, &reg, &err);
    CK(err);
    const size_t mbw2 = (size_t)net_width, cells = mbw2 * (size_t)(((height * pixel_rate) / 16) * 2);
    const size_t netbytes = cells * 4 > 0 ? cells * 4 : 4;
    cl_mem c = buf_in(cur, plane), sn = buf_in(src_net, netbytes), dn = buf_in(dst_net, netbytes);
    cl_kernel k = kern(g_prog_gpu, "luma_search_1step");
    ARG_MEM(k, 0, c); ARG_MEM(k, 1, refm); ARG_MEM(k, 2, sn); ARG_MEM(k, 3, dn);
    ARG_INT(k, 4, net_width); ARG_INT(k, 5, width); ARG_INT(k, 6, height); ARG_INT(k, 7, pixel_rate);
    run(k, round256((size_t)(width / 8) * (height / 8)), 256);     /* src/inter_part.h:110-122 */
    buf_out(dn, dst_net, netbytes);
    clReleaseMemObject(c); clReleaseMemObject(sn); clReleaseMemObject(refm); clReleaseMemObject(big);
}

void ref_luma_search_2step(const uint8_t *cur, const uint8_t *ref, const int16_t *net, int16_t *ref_net,
                           int32_t *ref_Bdiff, int width, int height) {
    need();
    const size_t plane = (size_t)width * height, b8 = plane / 64;
    cl_mem c = buf_in(cur, plane), n = buf_in(net, b8 * 4), rn = buf_in(ref_net, b8 * 4), bd = buf_in(ref_Bdiff, b8 * 4);
    image_buf img = image_in(ref, width, height);
    cl_kernel k = kern(g_prog_gpu_img, "luma_search_2step");
    ARG_MEM(k, 0, c); ARG_MEM(k, 1, img.pixels); ARG_MEM(k, 2, n); ARG_MEM(k, 3, rn); ARG_MEM(k, 4, bd);
    ARG_INT(k, 5, width); ARG_INT(k, 6, height);
    run(k, round256(b8), 256);                                       /* src/inter_part.h:201-223 */
    buf_out(rn, ref_net, b8 * 4);
    buf_out(bd, ref_Bdiff, b8 * 4);
    clReleaseMemObject(c); image_release(img); clReleaseMemObject(n);
}

void ref_select_reference(const int16_t *last_net, const int16_t *golden_net, const int16_t *altref_net,
                          const int32_t *last_Bdiff, const int32_t *golden_Bdiff, const int32_t *altref_Bdiff,
                          int32_t *MB_ref, int16_t *MB_vectors, int width, int height, int use_golden, int use_altref) {
    need();
    const size_t mbs = (size_t)(width / 16) * (height / 16), b8 = mbs * 4;
    cl_mem l = buf_in(last_net, b8 * 4), g = buf_in(golden_net, b8 * 4), a = buf_in(altref_net, b8 * 4);
    cl_mem lb = buf_in(last_Bdiff, b8 * 4), gb = buf_in(golden_Bdiff, b8 * 4), ab = buf_in(altref_Bdiff, b8 * 4);
    cl_mem r = buf_in(MB_ref, mbs * 4), v = buf_in(MB_vectors, mbs * 16);
    cl_kernel k = kern(g_prog_gpu, "select_reference");
    ARG_MEM(k, 0, l); ARG_MEM(k, 1, g); ARG_MEM(k, 2, a); ARG_MEM(k, 3, lb); ARG_MEM(k, 4, gb); ARG_MEM(k, 5, ab);
    ARG_MEM(k, 6, r); ARG_MEM(k, 7, v); ARG_INT(k, 8, width); ARG_INT(k, 9, use_golden); ARG_INT(k, 10, use_altref);
    run(k, mbs, 0);                                                  /* src/inter_part.h:251-254 */
    buf_out(r, MB_ref, mbs * 4);
    buf_out(v, MB_vectors, mbs * 16);
    clReleaseMemObject(l); clReleaseMemObject(g); clReleaseMemObject(a);
    clReleaseMemObject(lb); clReleaseMemObject(gb); clReleaseMemObject(ab);
}

void ref_pack_8x8_into_16x16(const int16_t *MB_vectors, int32_t *MB_parts, float *MB_SSIM, int mb_count) {
    need();
    const size_t mbs = (size_t)mb_count;
    cl_mem v = buf_in(MB_vectors, mbs * 16), p = buf_in(MB_parts, mbs * 4), s = buf_in(MB_SSIM, mbs * 4);
    cl_kernel k = kern(g_prog_gpu, "pack_8x8_into_16x16");
    ARG_MEM(k, 0, v); ARG_MEM(k, 1, p); ARG_MEM(k, 2, s);
    run(k, mbs, 0);                                                  /* src/inter_part.h:257-258 */
    buf_out(p, MB_parts, mbs * 4);
    buf_out(s, MB_SSIM, mbs * 4);
    clReleaseMemObject(v);
}

void ref_prepare_predictors_and_residual(const uint8_t *cur, const uint8_t *ref, uint8_t *predictor, int16_t *residual,
                                         const int32_t *MB_ref, const int16_t *MB_vectors, int width, int height,
                                         int plane, int ref_id) {
    need();
    const size_t px = (size_t)width * height;
    const int mbsz = plane ? 8 : 16;
    const size_t mbs = (size_t)(width / mbsz) * (height / mbsz);
    cl_mem c = buf_in(cur, px), pr = buf_in(predictor, px), rs = buf_in(residual, px * 2);
    cl_mem r = buf_in(MB_ref, mbs * 4), v = buf_in(MB_vectors, mbs * 16);
    image_buf img = image_in(ref, width, height);
    cl_kernel k = kern(g_prog_gpu_img, "prepare_predictors_and_residual");
    ARG_MEM(k, 0, c); ARG_MEM(k, 1, img.pixels); ARG_MEM(k, 2, pr); ARG_MEM(k, 3, rs); ARG_MEM(k, 4, r); ARG_MEM(k, 5, v);
    ARG_INT(k, 6, width); ARG_INT(k, 7, plane); ARG_INT(k, 8, ref_id);
    run(k, (size_t)(width / 4) * (height / 4), 0);                   /* src/inter_part.h:270-319 */
    buf_out(pr, predictor, px);
    buf_out(rs, residual, px * 2);
    clReleaseMemObject(c); image_release(img); clReleaseMemObject(r); clReleaseMemObject(v);
}

static size_t mbs_of_plane(int width, int height, int plane) {
    const int mbsz = plane ? 8 : 16;
    return (size_t)(width / mbsz) * (height / mbsz);
}

void ref_dct4x4(const int16_t *residual, int16_t *MB, int32_t *MB_segment_id, const int32_t *MB_parts,
                const float *MB_SSIM, int width, int height, const int32_t *SD, int segment_id, float SSIM_target, int plane) {
    need();
    const size_t px = (size_t)width * height, mbs = mbs_of_plane(width, height, plane);
    cl_mem rs = buf_in(residual, px * 2), mb = buf_in(MB, mbs * 800), sg = buf_in(MB_segment_id, mbs * 4);
    cl_mem pt = buf_in(MB_parts, mbs * 4), ss = buf_in(MB_SSIM, mbs * 4), sd = buf_in(SD, 44 * 4);
    cl_kernel k = kern(g_prog_gpu, "dct4x4");
    ARG_MEM(k, 0, rs); ARG_MEM(k, 1, mb); ARG_MEM(k, 2, sg); ARG_MEM(k, 3, pt); ARG_MEM(k, 4, ss); ARG_INT(k, 5, width);
    ARG_MEM(k, 6, sd); ARG_INT(k, 7, segment_id); ARG_FLT(k, 8, SSIM_target); ARG_INT(k, 9, plane);
    run(k, (size_t)(width / 4) * (height / 4), 0);                   /* src/inter_part.h:333-342 */
    buf_out(mb, MB, mbs * 800);
    buf_out(sg, MB_segment_id, mbs * 4);
    clReleaseMemObject(rs); clReleaseMemObject(pt); clReleaseMemObject(ss); clReleaseMemObject(sd);
}

void ref_wht4x4_iwht4x4(int16_t *MB, const int32_t *MB_segment_id, const int32_t *MB_parts, const int32_t *SD,
                        int segment_id, int mb_count) {
    need();
    const size_t mbs = (size_t)mb_count;
    cl_mem mb = buf_in(MB, mbs * 800), ss = buf_in(NULL, mbs * 4), sg = buf_in(MB_segment_id, mbs * 4), pt = buf_in(MB_parts, mbs * 4),
           sd = buf_in(SD, 44 * 4);
    cl_kernel k = kern(g_prog_gpu, "wht4x4_iwht4x4");
    ARG_MEM(k, 0, mb); ARG_MEM(k, 1, ss); ARG_MEM(k, 2, sg); ARG_MEM(k, 3, pt); ARG_MEM(k, 4, sd); ARG_INT(k, 5, segment_id);
    run(k, mbs, 0);                                                  /* src/inter_part.h:345-346 */
    buf_out(mb, MB, mbs * 800);
    clReleaseMemObject(ss); clReleaseMemObject(sg); clReleaseMemObject(pt); clReleaseMemObject(sd);
}

void ref_idct4x4(uint8_t *recon, const uint8_t *predictor, const int16_t *MB, const int32_t *MB_segment_id,
                 const int32_t *MB_parts, int width, int height, const int32_t *SD, int segment_id, int plane) {
    need();
    const size_t px = (size_t)width * height, mbs = mbs_of_plane(width, height, plane);
    cl_mem rc = buf_in(recon, px), pr = buf_in(predictor, px), mb = buf_in(MB, mbs * 800), sg = buf_in(MB_segment_id, mbs * 4);
    cl_mem pt = buf_in(MB_parts, mbs * 4), sd = buf_in(SD, 44 * 4);
    cl_kernel k = kern(g_prog_gpu, "idct4x4");
    ARG_MEM(k, 0, rc); ARG_MEM(k, 1, pr); ARG_MEM(k, 2, mb); ARG_MEM(k, 3, sg); ARG_MEM(k, 4, pt); ARG_INT(k, 5, width);
    ARG_MEM(k, 6, sd); ARG_INT(k, 7, segment_id); ARG_INT(k, 8, plane);
    run(k, (size_t)(width / 4) * (height / 4), 0);                   /* src/inter_part.h:349-357 */
    buf_out(rc, recon, px);
    clReleaseMemObject(pr); clReleaseMemObject(mb); clReleaseMemObject(sg); clReleaseMemObject(pt); clReleaseMemObject(sd);
}

void ref_count_SSIM(const uint8_t *f1, const uint8_t *f2, const int32_t *MB_segment_id, float *metric, int width,
                    int height, int segment_id, int mb_size) {
    need();
    const size_t px = (size_t)width * height, mbs = (size_t)(width / mb_size) * (height / mb_size);
    cl_mem a = buf_in(f1, px), b = buf_in(f2, px), sg = buf_in(MB_segment_id, mbs * 4), m = buf_in(metric, mbs * 4);
    cl_kernel k = kern(g_prog_gpu, mb_size == 16 ? "count_SSIM_luma" : "count_SSIM_chroma");
    ARG_MEM(k, 0, a); ARG_MEM(k, 1, b); ARG_MEM(k, 2, sg); ARG_MEM(k, 3, m); ARG_INT(k, 4, width); ARG_INT(k, 5, segment_id);
    run(k, mbs, 0);                                                  /* src/inter_part.h:361-369 */
    buf_out(m, metric, mbs * 4);
    clReleaseMemObject(a); clReleaseMemObject(b); clReleaseMemObject(sg);
}

void ref_gather_SSIM(const float *m1, const float *m2, const float *m3, float *MB_SSIM, int mb_count) {
    need();
    const size_t mbs = (size_t)mb_count;
    cl_mem a = buf_in(m1, mbs * 4), b = buf_in(m2, mbs * 4), c = buf_in(m3, mbs * 4), s = buf_in(MB_SSIM, mbs * 4);
    cl_kernel k = kern(g_prog_gpu, "gather_SSIM");
    ARG_MEM(k, 0, a); ARG_MEM(k, 1, b); ARG_MEM(k, 2, c); ARG_MEM(k, 3, s);
    run(k, mbs, 0);                                                  /* src/inter_part.h:376-377 */
    buf_out(s, MB_SSIM, mbs * 4);
    clReleaseMemObject(a); clReleaseMemObject(b); clReleaseMemObject(c);
}

/* ---- src/CPU_kernels.cl (the reference runs these on its CPU device; same source, same built-ins) --- */
void ref_prepare_filter_mask(const int16_t *MB, int32_t *MB_non_zero_coeffs, const int32_t *MB_parts, int32_t *mb_mask,
                             int width, int height) {
    need();
    const size_t mbs = (size_t)(width / 16) * (height / 16);
    cl_mem mb = buf_in(MB, mbs * 800), nz = buf_in(MB_non_zero_coeffs, mbs * 4), pt = buf_in(MB_parts, mbs * 4), mk = buf_in(mb_mask, mbs * 4);
    cl_kernel k = kern(g_prog_cpu, "prepare_filter_mask");
    ARG_MEM(k, 0, mb); ARG_MEM(k, 1, nz); ARG_MEM(k, 2, pt); ARG_MEM(k, 3, mk); ARG_INT(k, 4, width); ARG_INT(k, 5, height); ARG_INT(k, 6, 4);
    run(k, 4, 1);                                                    /* src/loop_filter.h:25-32 */
    buf_out(nz, MB_non_zero_coeffs, mbs * 4);
    buf_out(mk, mb_mask, mbs * 4);
    clReleaseMemObject(mb); clReleaseMemObject(pt);
}

void ref_loop_filter_frame(uint8_t *frame, const int32_t *MB_segment_ids, const int32_t *mb_mask, const int32_t *SD,
                           int width, int height, int mb_size) {
    need();
    const size_t px = (size_t)width * height, mbs = (size_t)(width / mb_size) * (height / mb_size);
    cl_mem fr = buf_in(frame, px), sg = buf_in(MB_segment_ids, mbs * 4), mk = buf_in(mb_mask, mbs * 4), sd = buf_in(SD, 44 * 4);
    cl_kernel k = kern(g_prog_cpu, mb_size == 16 ? "loop_filter_frame_luma" : "loop_filter_frame_chroma");
    ARG_MEM(k, 0, fr); ARG_MEM(k, 1, sg); ARG_MEM(k, 2, mk); ARG_MEM(k, 3, sd); ARG_INT(k, 4, width); ARG_INT(k, 5, height);
    run(k, 1, 1);                                                    /* src/loop_filter.h:143-170: one work-item per plane */
    buf_out(fr, frame, px);
    clReleaseMemObject(sg); clReleaseMemObject(mk); clReleaseMemObject(sd);
}

/* coefficient entropy stage, src/CPU_kernels.cl:347-778, one work-item per partition (src/vp8enc.cpp:48-94) */
#define PROBS_BYTES (4 * 8 * 3 * 11 * 4)
void ref_count_probs(const int16_t *MB, const int32_t *nz, const int32_t *MB_parts, uint32_t *coeff_probs,
                     uint32_t *coeff_probs_denom, uint8_t *third_context, int mb_height, int mb_width, int num_partitions) {
    need();
    const size_t mbs = (size_t)mb_height * mb_width, pb = (size_t)PROBS_BYTES * num_partitions;
    cl_mem mb = buf_in(MB, mbs * 800), n = buf_in(nz, mbs * 4), pt = buf_in(MB_parts, mbs * 4);
    cl_mem cp = buf_in(coeff_probs, pb), cd = buf_in(coeff_probs_denom, pb), tc = buf_in(third_context, mbs * 25);
    cl_kernel k = kern(g_prog_cpu, "count_probs");
    ARG_MEM(k, 0, mb); ARG_MEM(k, 1, n); ARG_MEM(k, 2, pt); ARG_MEM(k, 3, cp); ARG_MEM(k, 4, cd); ARG_MEM(k, 5, tc);
    ARG_INT(k, 6, mb_height); ARG_INT(k, 7, mb_width); ARG_INT(k, 8, num_partitions); ARG_INT(k, 9, 0);
    run(k, (size_t)num_partitions, 1);
    buf_out(cp, coeff_probs, pb);
    buf_out(cd, coeff_probs_denom, pb);
    buf_out(tc, third_context, mbs * 25);
    clReleaseMemObject(mb); clReleaseMemObject(n); clReleaseMemObject(pt);
}

void ref_num_div_denom(uint32_t *coeff_probs, const uint32_t *coeff_probs_denom, int num_partitions) {
    need();
    const size_t pb = (size_t)PROBS_BYTES * num_partitions;
    cl_mem cp = buf_in(coeff_probs, pb), cd = buf_in(coeff_probs_denom, pb);
    cl_kernel k = kern(g_prog_cpu, "num_div_denom");
    ARG_MEM(k, 0, cp); ARG_MEM(k, 1, cd); ARG_INT(k, 2, num_partitions);
    run(k, (size_t)num_partitions, 1);
    buf_out(cp, coeff_probs, pb);
    clReleaseMemObject(cd);
}

void ref_encode_coefficients(const int16_t *MB, const int32_t *nz, const int32_t *MB_parts, uint8_t *output,
                             int32_t *partition_sizes, const uint8_t *third_context, const uint32_t *coeff_probs,
                             int mb_height, int mb_width, int num_partitions, int partition_step) {
    need();
    const size_t mbs = (size_t)mb_height * mb_width, ob = (size_t)partition_step * num_partitions;
    cl_mem mb = buf_in(MB, mbs * 800), n = buf_in(nz, mbs * 4), pt = buf_in(MB_parts, mbs * 4), o = buf_in(output, ob);
    cl_mem ps = buf_in(partition_sizes, (size_t)num_partitions * 4), tc = buf_in(third_context, mbs * 25), cp = buf_in(coeff_probs, PROBS_BYTES);
    cl_kernel k = kern(g_prog_cpu, "encode_coefficients");
    ARG_MEM(k, 0, mb); ARG_MEM(k, 1, n); ARG_MEM(k, 2, pt); ARG_MEM(k, 3, o); ARG_MEM(k, 4, ps); ARG_MEM(k, 5, tc); ARG_MEM(k, 6, cp);
    ARG_INT(k, 7, mb_height); ARG_INT(k, 8, mb_width); ARG_INT(k, 9, num_partitions); ARG_INT(k, 10, partition_step);
    run(k, (size_t)num_partitions, 1);
    buf_out(o, output, ob);
    buf_out(ps, partition_sizes, (size_t)num_partitions * 4);
    clReleaseMemObject(mb); clReleaseMemObject(n); clReleaseMemObject(pt); clReleaseMemObject(tc); clReleaseMemObject(cp);
}
