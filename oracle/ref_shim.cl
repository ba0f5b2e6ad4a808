/*
 * ref_shim.cl -- OpenCL C built-ins needed to EXECUTE the reference's own kernels on x86.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/build_ref.sh compiles /root/reference/src/GPU_kernels.cl and
 * CPU_kernels.cl (from where they lie, nothing is copied into this repo) with clang's OpenCL C
 * front end for x86-64.  The resulting objects import the OpenCL C built-in functions by their
 * Itanium-mangled names; this file defines exactly those built-ins, each with the semantics the
 * OpenCL C 1.2 specification gives it (section 6.2.3 conversions, 6.12.3 integer functions,
 * 6.12.6 relational functions, 6.12.7 vload/vstore, 6.12.14 image reads).  It contains no part
 * of the encoder's algorithm.  
 */
/* Written in OpenCL C and compiled with the same front end and flags as the kernels (minus the
 * default header), so that small vectors such as uchar2 use the same calling convention. */
typedef __SIZE_TYPE__ size_t;
typedef unsigned char uchar;
typedef unsigned short ushort;
typedef unsigned int uint;
#define VEC(T, N) typedef T T##N __attribute__((ext_vector_type(N)))
VEC(uchar, 2); VEC(uchar, 4); VEC(uchar, 8); VEC(uchar, 16);
VEC(short, 2); VEC(short, 4); VEC(short, 8);
VEC(ushort, 8);
VEC(int, 2); VEC(int, 4); VEC(int, 16);
VEC(uint, 4); VEC(uint, 16);
VEC(float, 4);

/* ---- work-item functions: the driver sets the ids before every kernel call -------------- */
size_t ref_gid(void);  /* oracle/ref_driver.c */
size_t ref_lsz(void);
#define g_gid ref_gid()
#define g_lsz ref_lsz()

size_t b_get_global_id(uint d) __asm__("_Z13get_global_idj");
size_t b_get_global_id(uint d) { return d == 0 ? g_gid : 0; }
size_t b_get_local_id(uint d) __asm__("_Z12get_local_idj");
size_t b_get_local_id(uint d) { return d == 0 ? g_gid % g_lsz : 0; }
size_t b_get_local_size(uint d) __asm__("_Z14get_local_sizej");
size_t b_get_local_size(uint d) { return d == 0 ? g_lsz : 1; }

/* ---- conversions ------------------------------------------------------------------------ */
#define SAT8(v) ((v) < 0 ? 0 : ((v) > 255 ? 255 : (v)))
int2 b_cvt_int2_uc2(uchar2 v) __asm__("_Z12convert_int2Dv2_h");
int2 b_cvt_int2_uc2(uchar2 v) { return (int2)(v.x, v.y); }
int2 b_cvt_int2_s2(short2 v) __asm__("_Z12convert_int2Dv2_s");
int2 b_cvt_int2_s2(short2 v) { return (int2)(v.x, v.y); }
int4 b_cvt_int4_uc4(uchar4 v) __asm__("_Z12convert_int4Dv4_h");
int4 b_cvt_int4_uc4(uchar4 v) { return (int4)(v.x, v.y, v.z, v.w); }
int16 b_cvt_int16_uc16(uchar16 v) __asm__("_Z13convert_int16Dv16_h");
int16 b_cvt_int16_uc16(uchar16 v) { int16 r; for (int i = 0; i < 16; ++i) r[i] = v[i]; return r; }
int16 b_cvt_int16_ui16(uint16 v) __asm__("_Z13convert_int16Dv16_j");
int16 b_cvt_int16_ui16(uint16 v) { int16 r; for (int i = 0; i < 16; ++i) r[i] = (int)v[i]; return r; }
float4 b_cvt_float4_uc4(uchar4 v) __asm__("_Z14convert_float4Dv4_h");
float4 b_cvt_float4_uc4(uchar4 v) { return (float4)((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
short4 b_cvt_short4_i4(int4 v) __asm__("_Z14convert_short4Dv4_i");
short4 b_cvt_short4_i4(int4 v) { return (short4)((short)v.x, (short)v.y, (short)v.z, (short)v.w); }
short8 b_cvt_short8_uc8(uchar8 v) __asm__("_Z14convert_short8Dv8_h");
short8 b_cvt_short8_uc8(uchar8 v) { short8 r; for (int i = 0; i < 8; ++i) r[i] = v[i]; return r; }
uchar4 b_cvt_uc4_sat_i4(int4 v) __asm__("_Z18convert_uchar4_satDv4_i");
uchar4 b_cvt_uc4_sat_i4(int4 v) { uchar4 r; for (int i = 0; i < 4; ++i) r[i] = (uchar)SAT8(v[i]); return r; }
uchar8 b_cvt_uc8_sat_s8(short8 v) __asm__("_Z18convert_uchar8_satDv8_s");
uchar8 b_cvt_uc8_sat_s8(short8 v) { uchar8 r; for (int i = 0; i < 8; ++i) r[i] = (uchar)SAT8(v[i]); return r; }
uchar16 b_cvt_uc16_sat_i16(int16 v) __asm__("_Z19convert_uchar16_satDv16_i");
uchar16 b_cvt_uc16_sat_i16(int16 v) { uchar16 r; for (int i = 0; i < 16; ++i) r[i] = (uchar)SAT8(v[i]); return r; }

/* ---- integer functions: abs() returns the unsigned type ---------------------------------- */
uint b_abs_i(int v) __asm__("_Z3absi");
uint b_abs_i(int v) { return v < 0 ? 0u - (uint)v : (uint)v; }
ushort b_abs_s(short v) __asm__("_Z3abss");
ushort b_abs_s(short v) { return (ushort)(v < 0 ? -v : v); }
uint16 b_abs_i16(int16 v) __asm__("_Z3absDv16_i");
uint16 b_abs_i16(int16 v) { uint16 r; for (int i = 0; i < 16; ++i) r[i] = v[i] < 0 ? 0u - (uint)v[i] : (uint)v[i]; return r; }
ushort8 b_abs_s8(short8 v) __asm__("_Z3absDv8_s");
ushort8 b_abs_s8(short8 v) { ushort8 r; for (int i = 0; i < 8; ++i) r[i] = (ushort)(v[i] < 0 ? -v[i] : v[i]); return r; }
int b_mad24_i(int a, int b, int c) __asm__("_Z5mad24iii");
int b_mad24_i(int a, int b, int c) { return a * b + c; }
int2 b_mad24_i2(int2 a, int2 b, int2 c) __asm__("_Z5mad24Dv2_iS_S_");
int2 b_mad24_i2(int2 a, int2 b, int2 c) { return a * b + c; }

/* ---- mad(): a*b+c, product rounded (this file is built with -ffp-contract=off) ------------- */
float b_mad_f(float a, float b, float c) __asm__("_Z3madfff");
float b_mad_f(float a, float b, float c) { float p = a * b; return p + c; }
float4 b_mad_f4(float4 a, float4 b, float4 c) __asm__("_Z3madDv4_fS_S_");
float4 b_mad_f4(float4 a, float4 b, float4 c) { float4 p = a * b; return p + c; }

/* ---- select(a,b,c): scalar c != 0 ? b : a; vector: MSB of c ? b : a ------------------------ */
int b_select_i(int a, int b, int c) __asm__("_Z6selectiii");
int b_select_i(int a, int b, int c) { return c ? b : a; }
short b_select_s(short a, short b, short c) __asm__("_Z6selectsss");
short b_select_s(short a, short b, short c) { return c ? b : a; }
float b_select_f(float a, float b, int c) __asm__("_Z6selectffi");
float b_select_f(float a, float b, int c) { return c ? b : a; }
short8 b_select_s8(short8 a, short8 b, short8 c) __asm__("_Z6selectDv8_sS_S_");
short8 b_select_s8(short8 a, short8 b, short8 c) { short8 r; for (int i = 0; i < 8; ++i) r[i] = c[i] < 0 ? b[i] : a[i]; return r; }

/* ---- vload / vstore ---------------------------------------------------------------------- */
uchar2 b_vload2(size_t o, const __global uchar *p) __asm__("_Z6vload2mPU8CLglobalKh");
uchar2 b_vload2(size_t o, const __global uchar *p) { p += 2 * o; return (uchar2)(p[0], p[1]); }
uchar4 b_vload4(size_t o, const __global uchar *p) __asm__("_Z6vload4mPU8CLglobalKh");
uchar4 b_vload4(size_t o, const __global uchar *p) { p += 4 * o; return (uchar4)(p[0], p[1], p[2], p[3]); }
uchar8 b_vload8(size_t o, const __global uchar *p) __asm__("_Z6vload8mPU8CLglobalKh");
uchar8 b_vload8(size_t o, const __global uchar *p) { uchar8 r; p += 8 * o; for (int i = 0; i < 8; ++i) r[i] = p[i]; return r; }
void b_vstore4_uc(uchar4 v, size_t o, __global uchar *p) __asm__("_Z7vstore4Dv4_hmPU8CLglobalh");
void b_vstore4_uc(uchar4 v, size_t o, __global uchar *p) { p += 4 * o; for (int i = 0; i < 4; ++i) p[i] = v[i]; }
void b_vstore4_s(short4 v, size_t o, __global short *p) __asm__("_Z7vstore4Dv4_smPU8CLglobals");
void b_vstore4_s(short4 v, size_t o, __global short *p) { p += 4 * o; for (int i = 0; i < 4; ++i) p[i] = v[i]; }
void b_vstore8_uc(uchar8 v, size_t o, __global uchar *p) __asm__("_Z7vstore8Dv8_hmPU8CLglobalh");
void b_vstore8_uc(uchar8 v, size_t o, __global uchar *p) { p += 8 * o; for (int i = 0; i < 8; ++i) p[i] = v[i]; }

/* ---- images: CL_R / CL_UNSIGNED_INT8, unnormalised coords, CLK_ADDRESS_CLAMP_TO_EDGE ------- */
typedef struct { const __global uchar *data; int w, h; } ref_image;
__global void *b_translate_sampler(int v) __asm__("__translate_sampler_initializer");
__global void *b_translate_sampler(int v) { (void)v; return 0; }
uint4 b_read_imageui(const __global ref_image *img, __global void *smp, int2 c) __asm__("_Z12read_imageui14ocl_image2d_ro11ocl_samplerDv2_i");
uint4 b_read_imageui(const __global ref_image *img, __global void *smp, int2 c) {
    (void)smp;
    int x = c.x < 0 ? 0 : (c.x > img->w - 1 ? img->w - 1 : c.x);
    int y = c.y < 0 ? 0 : (c.y > img->h - 1 ? img->h - 1 : c.y);
    return (uint4)(img->data[(size_t)y * img->w + x], 0, 0, 1);
}
