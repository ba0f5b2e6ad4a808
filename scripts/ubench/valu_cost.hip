// valu_cost.hip -- issue cost, in SIMD cycles, of every integer VALU opcode class the hot kernels are made of.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost > profiles/valu_cost.json
//
// Saturated regime: every CU holds 8 waves per SIMD (2 workgroups of 1024 threads), each wave issues
// ITERS x 64 independent instructions of one opcode.  Wall time by hipEvents; the shader clock the chip actually held
// during the kernel from s_memtime / s_memrealtime inside it (MI355X_MICROARCH.md, DVFS give-back (6)).  Then
//   SIMD cycles per wave64 instruction = 1024 SIMDs x clock x wall / instructions.
// A second figure per opcode is in-wave: ticks of ONE wave's s_memtime per instruction with 8 waves per SIMD resident,
// divided by 8 -- it must agree (it does not depend on the hipEvent wall clock).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define REP16(x) x x x x x x x x x x x x x x x x
#define REGS : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)

#define OPS(X)                                                                                                                    \
    X(0, "v_add_u32", "v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n v_add_u32 %4, %4, %5\n v_add_u32 %6, %6, %7")               \
    X(1, "v_sub_u32", "v_sub_u32 %0, %0, %1\n v_sub_u32 %2, %2, %3\n v_sub_u32 %4, %4, %5\n v_sub_u32 %6, %6, %7")               \
    X(2, "v_xor_b32", "v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_xor_b32 %4, %4, %5\n v_xor_b32 %6, %6, %7")               \
    X(3, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %4, 3, %5\n v_lshlrev_b32 %6, 3, %7") \
    X(4, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %1\n v_ashrrev_i32 %2, 3, %3\n v_ashrrev_i32 %4, 3, %5\n v_ashrrev_i32 %6, 3, %7") \
    X(5, "v_mov_b32", "v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n v_mov_b32 %4, %5\n v_mov_b32 %6, %7")                                 \
    X(6, "v_cndmask_b32", "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %6, %6, %7, vcc") \
    X(7, "v_min_u32", "v_min_u32 %0, %0, %1\n v_min_u32 %2, %2, %3\n v_min_u32 %4, %4, %5\n v_min_u32 %6, %6, %7")               \
    X(8, "v_max_i32", "v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %3\n v_max_i32 %4, %4, %5\n v_max_i32 %6, %6, %7")               \
    X(9, "v_cmp_lt_i32", "v_cmp_lt_i32 vcc, %0, %1\n v_cmp_lt_i32 vcc, %2, %3\n v_cmp_lt_i32 vcc, %4, %5\n v_cmp_lt_i32 vcc, %6, %7") \
    X(10, "v_sub_u32_sdwa", "v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1\n v_sub_u32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_3\n v_sub_u32_sdwa %6, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n v_sub_u32_sdwa %7, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_2") \
    X(11, "v_add_u32_sdwa", "v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_add_u32_sdwa %3, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_add_u32_sdwa %6, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32_sdwa %7, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD") \
    X(12, "v_mov_b32_dpp", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_ror:4 row_mask:0xf bank_mask:0xf") \
    X(13, "v_add_u32_dpp", "v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %4, %5, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %6, %7, %6 row_ror:4 row_mask:0xf bank_mask:0xf") \
    X(14, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %3, %3, %4, %2\n v_add3_u32 %5, %5, %6, %2\n v_add3_u32 %7, %7, %1, %2") \
    X(15, "v_add_lshl_u32", "v_add_lshl_u32 %0, %0, %1, 3\n v_add_lshl_u32 %2, %2, %3, 3\n v_add_lshl_u32 %4, %4, %5, 3\n v_add_lshl_u32 %6, %6, %7, 3") \
    X(16, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 3, %1\n v_lshl_add_u32 %2, %2, 3, %3\n v_lshl_add_u32 %4, %4, 3, %5\n v_lshl_add_u32 %6, %6, 3, %7") \
    X(17, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 8, %1\n v_lshl_or_b32 %2, %2, 8, %3\n v_lshl_or_b32 %4, %4, 8, %5\n v_lshl_or_b32 %6, %6, 8, %7") \
    X(18, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %3, %3, %4, %2\n v_and_or_b32 %5, %5, %6, %2\n v_and_or_b32 %7, %7, %1, %2") \
    X(19, "v_bfe_u32", "v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %2, %3, 16, 8\n v_bfe_u32 %4, %5, 8, 8\n v_bfe_u32 %6, %7, 16, 8")    \
    X(20, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %3, %3, %4, %2\n v_perm_b32 %5, %5, %6, %2\n v_perm_b32 %7, %7, %1, %2") \
    X(21, "v_alignbyte_b32", "v_alignbyte_b32 %0, %0, %1, %2\n v_alignbyte_b32 %3, %3, %4, %2\n v_alignbyte_b32 %5, %5, %6, %2\n v_alignbyte_b32 %7, %7, %1, %2") \
    X(22, "v_sad_u16", "v_sad_u16 %0, %0, %1, %2\n v_sad_u16 %3, %3, %4, %2\n v_sad_u16 %5, %5, %6, %2\n v_sad_u16 %7, %7, %1, %2") \
    X(23, "v_sad_u8", "v_sad_u8 %0, %0, %1, %2\n v_sad_u8 %3, %3, %4, %2\n v_sad_u8 %5, %5, %6, %2\n v_sad_u8 %7, %7, %1, %2")  \
    X(24, "v_med3_i32", "v_med3_i32 %0, %0, %1, %2\n v_med3_i32 %3, %3, %4, %2\n v_med3_i32 %5, %5, %6, %2\n v_med3_i32 %7, %7, %1, %2") \
    X(25, "v_max3_i32", "v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %3, %3, %4, %2\n v_max3_i32 %5, %5, %6, %2\n v_max3_i32 %7, %7, %1, %2") \
    X(26, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %3, %3, %4, %2\n v_mad_u32_u24 %5, %5, %6, %2\n v_mad_u32_u24 %7, %7, %1, %2") \
    X(27, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %6, %6, %7") \
    X(28, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %6, %6, %7") \
    X(29, "v_mul_hi_u32", "v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %4, %4, %5\n v_mul_hi_u32 %6, %6, %7") \
    X(30, "v_dot2_i32_i16", "v_dot2_i32_i16 %0, %1, %2, %0\n v_dot2_i32_i16 %3, %4, %5, %3\n v_dot2_i32_i16 %6, %1, %5, %6\n v_dot2_i32_i16 %7, %4, %2, %7") \
    X(31, "v_dot2c_i32_i16", "v_dot2c_i32_i16 %0, %1, %2\n v_dot2c_i32_i16 %3, %4, %5\n v_dot2c_i32_i16 %6, %1, %5\n v_dot2c_i32_i16 %7, %4, %2") \
    X(32, "v_dot4_i32_i8", "v_dot4_i32_i8 %0, %1, %2, %0\n v_dot4_i32_i8 %3, %4, %5, %3\n v_dot4_i32_i8 %6, %1, %5, %6\n v_dot4_i32_i8 %7, %4, %2, %7") \
    X(33, "v_dot4c_i32_i8", "v_dot4c_i32_i8 %0, %1, %2\n v_dot4c_i32_i8 %3, %4, %5\n v_dot4c_i32_i8 %6, %1, %5\n v_dot4c_i32_i8 %7, %4, %2") \
    X(34, "v_dot4_u32_u8", "v_dot4_u32_u8 %0, %1, %2, %0\n v_dot4_u32_u8 %3, %4, %5, %3\n v_dot4_u32_u8 %6, %1, %5, %6\n v_dot4_u32_u8 %7, %4, %2, %7") \
    X(35, "v_pk_add_i16", "v_pk_add_i16 %0, %0, %1\n v_pk_add_i16 %2, %2, %3\n v_pk_add_i16 %4, %4, %5\n v_pk_add_i16 %6, %6, %7") \
    X(36, "v_pk_sub_i16", "v_pk_sub_i16 %0, %0, %1\n v_pk_sub_i16 %2, %2, %3\n v_pk_sub_i16 %4, %4, %5\n v_pk_sub_i16 %6, %6, %7") \
    X(37, "v_pk_ashrrev_i16", "v_pk_ashrrev_i16 %0, 4, %1\n v_pk_ashrrev_i16 %2, 4, %3\n v_pk_ashrrev_i16 %4, 4, %5\n v_pk_ashrrev_i16 %6, 4, %7") \
    X(38, "v_pk_lshlrev_b16", "v_pk_lshlrev_b16 %0, 3, %1\n v_pk_lshlrev_b16 %2, 3, %3\n v_pk_lshlrev_b16 %4, 3, %5\n v_pk_lshlrev_b16 %6, 3, %7") \
    X(39, "v_pk_max_i16", "v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %2, %2, %3\n v_pk_max_i16 %4, %4, %5\n v_pk_max_i16 %6, %6, %7") \
    X(40, "v_pk_mul_lo_u16", "v_pk_mul_lo_u16 %0, %0, %1\n v_pk_mul_lo_u16 %2, %2, %3\n v_pk_mul_lo_u16 %4, %4, %5\n v_pk_mul_lo_u16 %6, %6, %7") \
    X(41, "v_pk_mad_i16", "v_pk_mad_i16 %0, %0, %1, %2\n v_pk_mad_i16 %3, %3, %4, %2\n v_pk_mad_i16 %5, %5, %6, %2\n v_pk_mad_i16 %7, %7, %1, %2") \
    X(42, "v_ashr_pk_u8_i32", "v_ashr_pk_u8_i32 %0, %0, %1, 7\n v_ashr_pk_u8_i32 %2, %2, %3, 7\n v_ashr_pk_u8_i32 %4, %4, %5, 7\n v_ashr_pk_u8_i32 %6, %6, %7, 7") \
    X(43, "v_add_u16", "v_add_u16 %0, %0, %1\n v_add_u16 %2, %2, %3\n v_add_u16 %4, %4, %5\n v_add_u16 %6, %6, %7")             \
    X(44, "v_mad_i32_i16", "v_mad_i32_i16 %0, %0, %1, %2\n v_mad_i32_i16 %3, %3, %4, %2\n v_mad_i32_i16 %5, %5, %6, %2\n v_mad_i32_i16 %7, %7, %1, %2") \
    X(45, "v_sub_u32+v_sad mix 1:1", "v_sub_u32 %0, %0, %1\n v_sad_u16 %2, %2, %3, %1\n v_sub_u32 %4, %4, %5\n v_sad_u16 %6, %6, %7, %5") \
    X(46, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte0 %2, %3\n v_cvt_f32_ubyte0 %4, %5\n v_cvt_f32_ubyte0 %6, %7") \
    X(47, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %3, %3, %4, %2\n v_fma_f32 %5, %5, %6, %2\n v_fma_f32 %7, %7, %1, %2") \
    X(48, "v_and_b32", "v_and_b32 %0, %0, %1\n v_and_b32 %2, %2, %3\n v_and_b32 %4, %4, %5\n v_and_b32 %6, %6, %7") \
    X(49, "v_or_b32", "v_or_b32 %0, %0, %1\n v_or_b32 %2, %2, %3\n v_or_b32 %4, %4, %5\n v_or_b32 %6, %6, %7") \
    X(50, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %2, 3, %3\n v_lshrrev_b32 %4, 3, %5\n v_lshrrev_b32 %6, 3, %7") \
    X(51, "v_lshlrev_b32 by 1", "v_lshlrev_b32 %0, 1, %1\n v_lshlrev_b32 %2, 1, %3\n v_lshlrev_b32 %4, 1, %5\n v_lshlrev_b32 %6, 1, %7") \
    X(52, "v_lshlrev_b32 vgpr shift", "v_lshlrev_b32 %0, %0, %1\n v_lshlrev_b32 %2, %2, %3\n v_lshlrev_b32 %4, %4, %5\n v_lshlrev_b32 %6, %6, %7") \
    X(53, "v_ashrrev_i32 by 12", "v_ashrrev_i32 %0, 12, %1\n v_ashrrev_i32 %2, 12, %3\n v_ashrrev_i32 %4, 12, %5\n v_ashrrev_i32 %6, 12, %7") \
    X(54, "v_subrev_u32", "v_subrev_u32 %0, %0, %1\n v_subrev_u32 %2, %2, %3\n v_subrev_u32 %4, %4, %5\n v_subrev_u32 %6, %6, %7") \
    X(55, "v_add_u32 literal", "v_add_u32 %0, 0x12345, %1\n v_add_u32 %2, 0x12345, %3\n v_add_u32 %4, 0x12345, %5\n v_add_u32 %6, 0x12345, %7") \
    X(56, "v_add_u32 inline const", "v_add_u32 %0, 7, %1\n v_add_u32 %2, 7, %3\n v_add_u32 %4, 7, %5\n v_add_u32 %6, 7, %7") \
    X(57, "v_add_u32 sgpr", "v_add_u32 %0, s4, %1\n v_add_u32 %2, s4, %3\n v_add_u32 %4, s4, %5\n v_add_u32 %6, s4, %7") \
    X(58, "v_xor_b32 literal", "v_xor_b32 %0, 0x80008000, %1\n v_xor_b32 %2, 0x80008000, %3\n v_xor_b32 %4, 0x80008000, %5\n v_xor_b32 %6, 0x80008000, %7") \
    X(59, "v_sub_u16", "v_sub_u16 %0, %0, %1\n v_sub_u16 %2, %2, %3\n v_sub_u16 %4, %4, %5\n v_sub_u16 %6, %6, %7") \
    X(60, "v_max_u16", "v_max_u16 %0, %0, %1\n v_max_u16 %2, %2, %3\n v_max_u16 %4, %4, %5\n v_max_u16 %6, %6, %7") \
    X(61, "v_lshlrev_b16", "v_lshlrev_b16 %0, 3, %1\n v_lshlrev_b16 %2, 3, %3\n v_lshlrev_b16 %4, 3, %5\n v_lshlrev_b16 %6, 3, %7") \
    X(62, "v_mul_i32_i24", "v_mul_i32_i24 %0, %0, %1\n v_mul_i32_i24 %2, %2, %3\n v_mul_i32_i24 %4, %4, %5\n v_mul_i32_i24 %6, %6, %7") \
    X(63, "v_not_b32", "v_not_b32 %0, %1\n v_not_b32 %2, %3\n v_not_b32 %4, %5\n v_not_b32 %6, %7") \
    X(64, "v_min_i32", "v_min_i32 %0, %0, %1\n v_min_i32 %2, %2, %3\n v_min_i32 %4, %4, %5\n v_min_i32 %6, %6, %7") \
    X(65, "v_add_co_u32", "v_add_co_u32 %0, vcc, %0, %1\n v_add_co_u32 %2, vcc, %2, %3\n v_add_co_u32 %4, vcc, %4, %5\n v_add_co_u32 %6, vcc, %6, %7") \
    X(66, "v_add_f32", "v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n v_add_f32 %4, %4, %5\n v_add_f32 %6, %6, %7") \
    X(67, "v_mul_f32", "v_mul_f32 %0, %0, %1\n v_mul_f32 %2, %2, %3\n v_mul_f32 %4, %4, %5\n v_mul_f32 %6, %6, %7") \
    X(68, "v_max_f32", "v_max_f32 %0, %0, %1\n v_max_f32 %2, %2, %3\n v_max_f32 %4, %4, %5\n v_max_f32 %6, %6, %7") \
    X(69, "v_fmac_f32", "v_fmac_f32 %0, %0, %1\n v_fmac_f32 %2, %2, %3\n v_fmac_f32 %4, %4, %5\n v_fmac_f32 %6, %6, %7") \
    X(70, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %1\n v_cvt_f32_i32 %2, %3\n v_cvt_f32_i32 %4, %5\n v_cvt_f32_i32 %6, %7") \
    X(71, "v_cndmask_b32 e64 sgpr pair", "v_cndmask_b32 %0, %0, %1, s[6:7]\n v_cndmask_b32 %2, %2, %3, s[6:7]\n v_cndmask_b32 %4, %4, %5, s[6:7]\n v_cndmask_b32 %6, %6, %7, s[6:7]") \
    X(72, "v_cmp+v_cndmask vcc pair", "v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_i32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc") \
    X(73, "v_sat_pk_u8_i16", "v_sat_pk_u8_i16 %0, %1\n v_sat_pk_u8_i16 %2, %3\n v_sat_pk_u8_i16 %4, %5\n v_sat_pk_u8_i16 %6, %7") \
    X(74, "v_pk_sub_u16", "v_pk_sub_u16 %0, %0, %1\n v_pk_sub_u16 %2, %2, %3\n v_pk_sub_u16 %4, %4, %5\n v_pk_sub_u16 %6, %6, %7") \
    X(75, "v_pk_min_u16", "v_pk_min_u16 %0, %0, %1\n v_pk_min_u16 %2, %2, %3\n v_pk_min_u16 %4, %4, %5\n v_pk_min_u16 %6, %6, %7") \
    X(76, "v_pk_add_u16 op_sel", "v_pk_add_u16 %0, %0, %1 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_add_u16 %2, %2, %3 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_add_u16 %4, %4, %5 op_sel:[1,0] op_sel_hi:[0,1]\n v_pk_add_u16 %6, %6, %7 op_sel:[1,0] op_sel_hi:[0,1]") \
    X(77, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %0\n v_bfi_b32 %2, %2, %3, %2\n v_bfi_b32 %4, %4, %5, %4\n v_bfi_b32 %6, %6, %7, %6") \
    X(78, "v_xad_u32", "v_xad_u32 %0, %0, %1, %0\n v_xad_u32 %2, %2, %3, %2\n v_xad_u32 %4, %4, %5, %4\n v_xad_u32 %6, %6, %7, %6") \
    X(79, "v_mad_u32_u16", "v_mad_u32_u16 %0, %0, %1, %0\n v_mad_u32_u16 %2, %2, %3, %2\n v_mad_u32_u16 %4, %4, %5, %4\n v_mad_u32_u16 %6, %6, %7, %6") \
    X(80, "v_mad_u16", "v_mad_u16 %0, %0, %1, %0\n v_mad_u16 %2, %2, %3, %2\n v_mad_u16 %4, %4, %5, %4\n v_mad_u16 %6, %6, %7, %6") \
    X(81, "v_msad_u8", "v_msad_u8 %0, %0, %1, %0\n v_msad_u8 %2, %2, %3, %2\n v_msad_u8 %4, %4, %5, %4\n v_msad_u8 %6, %6, %7, %6") \
    X(82, "v_lerp_u8", "v_lerp_u8 %0, %0, %1, %0\n v_lerp_u8 %2, %2, %3, %2\n v_lerp_u8 %4, %4, %5, %4\n v_lerp_u8 %6, %6, %7, %6") \
    X(83, "v_cvt_pk_u8_f32", "v_cvt_pk_u8_f32 %0, %0, %1, %0\n v_cvt_pk_u8_f32 %2, %2, %3, %2\n v_cvt_pk_u8_f32 %4, %4, %5, %4\n v_cvt_pk_u8_f32 %6, %6, %7, %6")

constexpr int ITERS = 2000;   // long enough that the ramp of dispatching 8192 waves does not show

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, unsigned long long *stamps) {
    uint32_t a = threadIdx.x * 7 + 1, b = threadIdx.x * 13 + 5, c = threadIdx.x ^ 0x55, d = threadIdx.x + 99;
    uint32_t e = a + 1, f = b + 2, g = c + 3, h = d + 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; ++i) {
#define X(N, NAME, ASM) if (OP == N) { REP16(asm volatile(ASM REGS :: "vcc", "s4", "s6", "s7");) }
        OPS(X)
#undef X
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 16 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int OP>
void run(const char *name, bool last) {
    static uint32_t *out = nullptr;
    static unsigned long long *stamps = nullptr, *h = nullptr;
    const int blocks = 512, waves = blocks * 16;
    if (!out) {
        hipMalloc(&out, (size_t)blocks * 1024 * 4);
        hipMalloc(&stamps, waves * 16);
        h = (unsigned long long *)malloc(waves * 16);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, out, stamps);   // warm
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, out, stamps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, stamps, waves * 16, hipMemcpyDeviceToHost);
    double ticks = 0, real = 0;
    for (int i = 0; i < waves; ++i) { ticks += h[2 * i]; real += h[2 * i + 1]; }
    const double clock_ghz = ticks / real * 0.1;                // s_memrealtime runs at 100 MHz
    const double instr = (double)waves * ITERS * 64;            // wave64 instructions
    const double cyc_wall = 1024.0 * clock_ghz * ms * 1e6 / instr;
    const double cyc_wave = ticks / waves / (ITERS * 64.0) / 8.0;   // 8 waves share a SIMD
    printf("  \"%s\": {\"simd_cycles\": %.3f, \"simd_cycles_in_wave\": %.3f, \"clock_ghz\": %.3f, \"winstr_per_ns\": %.1f}%s\n", name, cyc_wall,
           cyc_wave, clock_ghz, instr / (ms * 1e6), last ? "" : ",");
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("{\n \"device\": \"%s\", \"cus\": %d, \"regime\": \"8 waves per SIMD on every CU, %d x 64 independent instructions per wave; simd_cycles = 1024 SIMDs x "
           "in-kernel clock x hipEvent wall / wave64 instructions; simd_cycles_in_wave = one wave's s_memtime ticks per instruction / 8\",\n \"ops\": {\n",
           p.gcnArchName, p.multiProcessorCount, ITERS);
#define X(N, NAME, ASM) run<N>(NAME, N == 83);
    OPS(X)
#undef X
    printf(" }\n}\n");
    return 0;
}
