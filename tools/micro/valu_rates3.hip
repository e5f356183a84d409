// Issue cost of 16 back-to-back independent instructions in ONE asm block (no compiler-inserted nops).  8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define R16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define OPS(X) \
    X(0, "v_cndmask_b32 vcc (VOP2)", "v_cndmask_b32 %0, %0, %16, vcc\nv_cndmask_b32 %1, %1, %16, vcc\nv_cndmask_b32 %2, %2, %16, vcc\nv_cndmask_b32 %3, %3, %16, vcc\nv_cndmask_b32 %4, %4, %16, vcc\nv_cndmask_b32 %5, %5, %16, vcc\nv_cndmask_b32 %6, %6, %16, vcc\nv_cndmask_b32 %7, %7, %16, vcc\nv_cndmask_b32 %8, %8, %16, vcc\nv_cndmask_b32 %9, %9, %16, vcc\nv_cndmask_b32 %10, %10, %16, vcc\nv_cndmask_b32 %11, %11, %16, vcc\nv_cndmask_b32 %12, %12, %16, vcc\nv_cndmask_b32 %13, %13, %16, vcc\nv_cndmask_b32 %14, %14, %16, vcc\nv_cndmask_b32 %15, %15, %16, vcc\n") \
    X(1, "v_cndmask_b32_e64 s[10:11]", "v_cndmask_b32_e64 %0, %0, %16, s[10:11]\nv_cndmask_b32_e64 %1, %1, %16, s[10:11]\nv_cndmask_b32_e64 %2, %2, %16, s[10:11]\nv_cndmask_b32_e64 %3, %3, %16, s[10:11]\nv_cndmask_b32_e64 %4, %4, %16, s[10:11]\nv_cndmask_b32_e64 %5, %5, %16, s[10:11]\nv_cndmask_b32_e64 %6, %6, %16, s[10:11]\nv_cndmask_b32_e64 %7, %7, %16, s[10:11]\nv_cndmask_b32_e64 %8, %8, %16, s[10:11]\nv_cndmask_b32_e64 %9, %9, %16, s[10:11]\nv_cndmask_b32_e64 %10, %10, %16, s[10:11]\nv_cndmask_b32_e64 %11, %11, %16, s[10:11]\nv_cndmask_b32_e64 %12, %12, %16, s[10:11]\nv_cndmask_b32_e64 %13, %13, %16, s[10:11]\nv_cndmask_b32_e64 %14, %14, %16, s[10:11]\nv_cndmask_b32_e64 %15, %15, %16, s[10:11]\n") \
    X(2, "v_cmp_gt_u32 vcc", "v_cmp_gt_u32 vcc, %16, %0\nv_cmp_gt_u32 vcc, %16, %1\nv_cmp_gt_u32 vcc, %16, %2\nv_cmp_gt_u32 vcc, %16, %3\nv_cmp_gt_u32 vcc, %16, %4\nv_cmp_gt_u32 vcc, %16, %5\nv_cmp_gt_u32 vcc, %16, %6\nv_cmp_gt_u32 vcc, %16, %7\nv_cmp_gt_u32 vcc, %16, %8\nv_cmp_gt_u32 vcc, %16, %9\nv_cmp_gt_u32 vcc, %16, %10\nv_cmp_gt_u32 vcc, %16, %11\nv_cmp_gt_u32 vcc, %16, %12\nv_cmp_gt_u32 vcc, %16, %13\nv_cmp_gt_u32 vcc, %16, %14\nv_cmp_gt_u32 vcc, %16, %15\n") \
    X(3, "v_cmp_gt_u32_e64 s[10:11]", "v_cmp_gt_u32_e64 s[10:11], %16, %0\nv_cmp_gt_u32_e64 s[10:11], %16, %1\nv_cmp_gt_u32_e64 s[10:11], %16, %2\nv_cmp_gt_u32_e64 s[10:11], %16, %3\nv_cmp_gt_u32_e64 s[10:11], %16, %4\nv_cmp_gt_u32_e64 s[10:11], %16, %5\nv_cmp_gt_u32_e64 s[10:11], %16, %6\nv_cmp_gt_u32_e64 s[10:11], %16, %7\nv_cmp_gt_u32_e64 s[10:11], %16, %8\nv_cmp_gt_u32_e64 s[10:11], %16, %9\nv_cmp_gt_u32_e64 s[10:11], %16, %10\nv_cmp_gt_u32_e64 s[10:11], %16, %11\nv_cmp_gt_u32_e64 s[10:11], %16, %12\nv_cmp_gt_u32_e64 s[10:11], %16, %13\nv_cmp_gt_u32_e64 s[10:11], %16, %14\nv_cmp_gt_u32_e64 s[10:11], %16, %15\n") \
    X(4, "v_sub_u32", "v_sub_u32 %0, %0, %16\nv_sub_u32 %1, %1, %16\nv_sub_u32 %2, %2, %16\nv_sub_u32 %3, %3, %16\nv_sub_u32 %4, %4, %16\nv_sub_u32 %5, %5, %16\nv_sub_u32 %6, %6, %16\nv_sub_u32 %7, %7, %16\nv_sub_u32 %8, %8, %16\nv_sub_u32 %9, %9, %16\nv_sub_u32 %10, %10, %16\nv_sub_u32 %11, %11, %16\nv_sub_u32 %12, %12, %16\nv_sub_u32 %13, %13, %16\nv_sub_u32 %14, %14, %16\nv_sub_u32 %15, %15, %16\n") \
    X(5, "v_or_b32", "v_or_b32 %0, %16, %0\nv_or_b32 %1, %16, %1\nv_or_b32 %2, %16, %2\nv_or_b32 %3, %16, %3\nv_or_b32 %4, %16, %4\nv_or_b32 %5, %16, %5\nv_or_b32 %6, %16, %6\nv_or_b32 %7, %16, %7\nv_or_b32 %8, %16, %8\nv_or_b32 %9, %16, %9\nv_or_b32 %10, %16, %10\nv_or_b32 %11, %16, %11\nv_or_b32 %12, %16, %12\nv_or_b32 %13, %16, %13\nv_or_b32 %14, %16, %14\nv_or_b32 %15, %16, %15\n") \
    X(6, "v_xor_b32", "v_xor_b32 %0, %16, %0\nv_xor_b32 %1, %16, %1\nv_xor_b32 %2, %16, %2\nv_xor_b32 %3, %16, %3\nv_xor_b32 %4, %16, %4\nv_xor_b32 %5, %16, %5\nv_xor_b32 %6, %16, %6\nv_xor_b32 %7, %16, %7\nv_xor_b32 %8, %16, %8\nv_xor_b32 %9, %16, %9\nv_xor_b32 %10, %16, %10\nv_xor_b32 %11, %16, %11\nv_xor_b32 %12, %16, %12\nv_xor_b32 %13, %16, %13\nv_xor_b32 %14, %16, %14\nv_xor_b32 %15, %16, %15\n") \
    X(7, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %0\nv_ashrrev_i32 %1, 3, %1\nv_ashrrev_i32 %2, 3, %2\nv_ashrrev_i32 %3, 3, %3\nv_ashrrev_i32 %4, 3, %4\nv_ashrrev_i32 %5, 3, %5\nv_ashrrev_i32 %6, 3, %6\nv_ashrrev_i32 %7, 3, %7\nv_ashrrev_i32 %8, 3, %8\nv_ashrrev_i32 %9, 3, %9\nv_ashrrev_i32 %10, 3, %10\nv_ashrrev_i32 %11, 3, %11\nv_ashrrev_i32 %12, 3, %12\nv_ashrrev_i32 %13, 3, %13\nv_ashrrev_i32 %14, 3, %14\nv_ashrrev_i32 %15, 3, %15\n") \
    X(8, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0\nv_lshlrev_b32 %1, 3, %1\nv_lshlrev_b32 %2, 3, %2\nv_lshlrev_b32 %3, 3, %3\nv_lshlrev_b32 %4, 3, %4\nv_lshlrev_b32 %5, 3, %5\nv_lshlrev_b32 %6, 3, %6\nv_lshlrev_b32 %7, 3, %7\nv_lshlrev_b32 %8, 3, %8\nv_lshlrev_b32 %9, 3, %9\nv_lshlrev_b32 %10, 3, %10\nv_lshlrev_b32 %11, 3, %11\nv_lshlrev_b32 %12, 3, %12\nv_lshlrev_b32 %13, 3, %13\nv_lshlrev_b32 %14, 3, %14\nv_lshlrev_b32 %15, 3, %15\n") \
    X(9, "v_add_u32_dpp row_shr:1", "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %8, %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %9, %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %10, %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %11, %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %12, %12, %12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %13, %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %14, %14, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_add_u32_dpp %15, %15, %15 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n") \
    X(10, "v_mov_b32_dpp row_shr:1", "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %8, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %9, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %10, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %11, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %12, %12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %13, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %14, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\nv_mov_b32_dpp %15, %15 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n") \
    X(11, "v_min_u32", "v_min_u32 %0, %16, %0\nv_min_u32 %1, %16, %1\nv_min_u32 %2, %16, %2\nv_min_u32 %3, %16, %3\nv_min_u32 %4, %16, %4\nv_min_u32 %5, %16, %5\nv_min_u32 %6, %16, %6\nv_min_u32 %7, %16, %7\nv_min_u32 %8, %16, %8\nv_min_u32 %9, %16, %9\nv_min_u32 %10, %16, %10\nv_min_u32 %11, %16, %11\nv_min_u32 %12, %16, %12\nv_min_u32 %13, %16, %13\nv_min_u32 %14, %16, %14\nv_min_u32 %15, %16, %15\n") \
    X(12, "v_max_u32", "v_max_u32 %0, %16, %0\nv_max_u32 %1, %16, %1\nv_max_u32 %2, %16, %2\nv_max_u32 %3, %16, %3\nv_max_u32 %4, %16, %4\nv_max_u32 %5, %16, %5\nv_max_u32 %6, %16, %6\nv_max_u32 %7, %16, %7\nv_max_u32 %8, %16, %8\nv_max_u32 %9, %16, %9\nv_max_u32 %10, %16, %10\nv_max_u32 %11, %16, %11\nv_max_u32 %12, %16, %12\nv_max_u32 %13, %16, %13\nv_max_u32 %14, %16, %14\nv_max_u32 %15, %16, %15\n") \
    X(13, "v_msad_u8", "v_msad_u8 %0, %16, %16, %0\nv_msad_u8 %1, %16, %16, %1\nv_msad_u8 %2, %16, %16, %2\nv_msad_u8 %3, %16, %16, %3\nv_msad_u8 %4, %16, %16, %4\nv_msad_u8 %5, %16, %16, %5\nv_msad_u8 %6, %16, %16, %6\nv_msad_u8 %7, %16, %16, %7\nv_msad_u8 %8, %16, %16, %8\nv_msad_u8 %9, %16, %16, %9\nv_msad_u8 %10, %16, %16, %10\nv_msad_u8 %11, %16, %16, %11\nv_msad_u8 %12, %16, %16, %12\nv_msad_u8 %13, %16, %16, %13\nv_msad_u8 %14, %16, %16, %14\nv_msad_u8 %15, %16, %16, %15\n") \
    X(14, "v_add_u32 (sgpr src0)", "v_add_u32 %0, s10, %0\nv_add_u32 %1, s10, %1\nv_add_u32 %2, s10, %2\nv_add_u32 %3, s10, %3\nv_add_u32 %4, s10, %4\nv_add_u32 %5, s10, %5\nv_add_u32 %6, s10, %6\nv_add_u32 %7, s10, %7\nv_add_u32 %8, s10, %8\nv_add_u32 %9, s10, %9\nv_add_u32 %10, s10, %10\nv_add_u32 %11, s10, %11\nv_add_u32 %12, s10, %12\nv_add_u32 %13, s10, %13\nv_add_u32 %14, s10, %14\nv_add_u32 %15, s10, %15\n") \
    X(15, "v_and_or_b32", "v_and_or_b32 %0, %0, %16, %16\nv_and_or_b32 %1, %1, %16, %16\nv_and_or_b32 %2, %2, %16, %16\nv_and_or_b32 %3, %3, %16, %16\nv_and_or_b32 %4, %4, %16, %16\nv_and_or_b32 %5, %5, %16, %16\nv_and_or_b32 %6, %6, %16, %16\nv_and_or_b32 %7, %7, %16, %16\nv_and_or_b32 %8, %8, %16, %16\nv_and_or_b32 %9, %9, %16, %16\nv_and_or_b32 %10, %10, %16, %16\nv_and_or_b32 %11, %11, %16, %16\nv_and_or_b32 %12, %12, %16, %16\nv_and_or_b32 %13, %13, %16, %16\nv_and_or_b32 %14, %14, %16, %16\nv_and_or_b32 %15, %15, %16, %16\n") \
    X(16, "v_bfe_u32", "v_bfe_u32 %0, %0, 16, 16\nv_bfe_u32 %1, %1, 16, 16\nv_bfe_u32 %2, %2, 16, 16\nv_bfe_u32 %3, %3, 16, 16\nv_bfe_u32 %4, %4, 16, 16\nv_bfe_u32 %5, %5, 16, 16\nv_bfe_u32 %6, %6, 16, 16\nv_bfe_u32 %7, %7, 16, 16\nv_bfe_u32 %8, %8, 16, 16\nv_bfe_u32 %9, %9, 16, 16\nv_bfe_u32 %10, %10, 16, 16\nv_bfe_u32 %11, %11, 16, 16\nv_bfe_u32 %12, %12, 16, 16\nv_bfe_u32 %13, %13, 16, 16\nv_bfe_u32 %14, %14, 16, 16\nv_bfe_u32 %15, %15, 16, 16\n") \
    X(17, "v_perm_b32", "v_perm_b32 %0, %0, %16, %16\nv_perm_b32 %1, %1, %16, %16\nv_perm_b32 %2, %2, %16, %16\nv_perm_b32 %3, %3, %16, %16\nv_perm_b32 %4, %4, %16, %16\nv_perm_b32 %5, %5, %16, %16\nv_perm_b32 %6, %6, %16, %16\nv_perm_b32 %7, %7, %16, %16\nv_perm_b32 %8, %8, %16, %16\nv_perm_b32 %9, %9, %16, %16\nv_perm_b32 %10, %10, %16, %16\nv_perm_b32 %11, %11, %16, %16\nv_perm_b32 %12, %12, %16, %16\nv_perm_b32 %13, %13, %16, %16\nv_perm_b32 %14, %14, %16, %16\nv_perm_b32 %15, %15, %16, %16\n") \
    X(18, "v_add_u32 x16 (reference)", "v_add_u32 %0, %16, %0\nv_add_u32 %1, %16, %1\nv_add_u32 %2, %16, %2\nv_add_u32 %3, %16, %3\nv_add_u32 %4, %16, %4\nv_add_u32 %5, %16, %5\nv_add_u32 %6, %16, %6\nv_add_u32 %7, %16, %7\nv_add_u32 %8, %16, %8\nv_add_u32 %9, %16, %9\nv_add_u32 %10, %16, %10\nv_add_u32 %11, %16, %11\nv_add_u32 %12, %16, %12\nv_add_u32 %13, %16, %13\nv_add_u32 %14, %16, %14\nv_add_u32 %15, %16, %15\n") \
    X(19, "v_subrev_u32", "v_subrev_u32 %0, %16, %0\nv_subrev_u32 %1, %16, %1\nv_subrev_u32 %2, %16, %2\nv_subrev_u32 %3, %16, %3\nv_subrev_u32 %4, %16, %4\nv_subrev_u32 %5, %16, %5\nv_subrev_u32 %6, %16, %6\nv_subrev_u32 %7, %16, %7\nv_subrev_u32 %8, %16, %8\nv_subrev_u32 %9, %16, %9\nv_subrev_u32 %10, %16, %10\nv_subrev_u32 %11, %16, %11\nv_subrev_u32 %12, %16, %12\nv_subrev_u32 %13, %16, %13\nv_subrev_u32 %14, %16, %14\nv_subrev_u32 %15, %16, %15\n") \
    X(20, "v_mul_u32_u24", "v_mul_u32_u24 %0, %16, %0\nv_mul_u32_u24 %1, %16, %1\nv_mul_u32_u24 %2, %16, %2\nv_mul_u32_u24 %3, %16, %3\nv_mul_u32_u24 %4, %16, %4\nv_mul_u32_u24 %5, %16, %5\nv_mul_u32_u24 %6, %16, %6\nv_mul_u32_u24 %7, %16, %7\nv_mul_u32_u24 %8, %16, %8\nv_mul_u32_u24 %9, %16, %9\nv_mul_u32_u24 %10, %16, %10\nv_mul_u32_u24 %11, %16, %11\nv_mul_u32_u24 %12, %16, %12\nv_mul_u32_u24 %13, %16, %13\nv_mul_u32_u24 %14, %16, %14\nv_mul_u32_u24 %15, %16, %15\n") \
    X(21, "v_mad_u32_u24", "v_mad_u32_u24 %0, %16, %16, %0\nv_mad_u32_u24 %1, %16, %16, %1\nv_mad_u32_u24 %2, %16, %16, %2\nv_mad_u32_u24 %3, %16, %16, %3\nv_mad_u32_u24 %4, %16, %16, %4\nv_mad_u32_u24 %5, %16, %16, %5\nv_mad_u32_u24 %6, %16, %16, %6\nv_mad_u32_u24 %7, %16, %16, %7\nv_mad_u32_u24 %8, %16, %16, %8\nv_mad_u32_u24 %9, %16, %16, %9\nv_mad_u32_u24 %10, %16, %16, %10\nv_mad_u32_u24 %11, %16, %16, %11\nv_mad_u32_u24 %12, %16, %16, %12\nv_mad_u32_u24 %13, %16, %16, %13\nv_mad_u32_u24 %14, %16, %16, %14\nv_mad_u32_u24 %15, %16, %16, %15\n") \
    X(22, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %16\nv_lshl_or_b32 %1, %1, 3, %16\nv_lshl_or_b32 %2, %2, 3, %16\nv_lshl_or_b32 %3, %3, 3, %16\nv_lshl_or_b32 %4, %4, 3, %16\nv_lshl_or_b32 %5, %5, 3, %16\nv_lshl_or_b32 %6, %6, 3, %16\nv_lshl_or_b32 %7, %7, 3, %16\nv_lshl_or_b32 %8, %8, 3, %16\nv_lshl_or_b32 %9, %9, 3, %16\nv_lshl_or_b32 %10, %10, 3, %16\nv_lshl_or_b32 %11, %11, 3, %16\nv_lshl_or_b32 %12, %12, 3, %16\nv_lshl_or_b32 %13, %13, 3, %16\nv_lshl_or_b32 %14, %14, 3, %16\nv_lshl_or_b32 %15, %15, 3, %16\n") \
    X(23, "v_sad_u16", "v_sad_u16 %0, %16, %16, %0\nv_sad_u16 %1, %16, %16, %1\nv_sad_u16 %2, %16, %16, %2\nv_sad_u16 %3, %16, %16, %3\nv_sad_u16 %4, %16, %16, %4\nv_sad_u16 %5, %16, %16, %5\nv_sad_u16 %6, %16, %16, %6\nv_sad_u16 %7, %16, %16, %7\nv_sad_u16 %8, %16, %16, %8\nv_sad_u16 %9, %16, %16, %9\nv_sad_u16 %10, %16, %16, %10\nv_sad_u16 %11, %16, %16, %11\nv_sad_u16 %12, %16, %16, %12\nv_sad_u16 %13, %16, %16, %13\nv_sad_u16 %14, %16, %16, %14\nv_sad_u16 %15, %16, %16, %15\n")

template <int OP>
__global__ void __launch_bounds__(256) probe(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed * (i + 1) + threadIdx.x;
    uint32_t b = seed ^ 0x01020304u;
    asm volatile("v_cmp_gt_u32 vcc, %0, %1\n s_mov_b64 s[10:11], vcc" : : "v"(b), "v"(r[0]) : "vcc", "s10", "s11");
    for (int it = 0; it < iters; it++) {
#define OPN(N, t) t
#define X(N, NAME, ASM) \
        if (OP == N) asm volatile(ASM : \
            "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), \
            "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : "v"(b) : "vcc", "s10", "s11");
        OPS(X)
#undef X
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s ^= r[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char* name, uint32_t* out)
{
    const int iters = 20000, wps = 8, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, 2000, 12345u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-30s %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wps));
}

int main()
{
    uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4);
#define X(N, NAME, ASM) run<N>(NAME, out);
    OPS(X)
#undef X
    return 0;
}
