// tools/valu_ubench.hip -- VALU issue-rate microbenchmark for gfx950 (measurement aid, not product).
// For each instruction kind: 8 independent register chains, 4096 instructions per lane-loop, launched
// with W waves per SIMD.  Prints wave-instructions per cycle per CU and the implied cycles per
// wave64 instruction per SIMD.  Used to price the per-cell instruction mix of the Viterbi kernel
// (DESIGN.md section 5).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
#define BODY(OP)                                                                                   \
  asm volatile(REP8(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n"    \
                    OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n")   \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
               : "v"(b));

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b = seed * 0.5f + threadIdx.x * 1e-9f;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { BODY("v_add_f32") }
    if (KIND == 1) { BODY("v_mul_f32") }
    if (KIND == 2) { BODY("v_max_f32") }
    if (KIND == 3) { BODY("v_fma_f32 %0, %0, %8, %0 ;") }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// packed fp32: 64-bit register pairs
typedef float float2v __attribute__((ext_vector_type(2)));
#define BODYPK(OP)                                                                                 \
  asm volatile(REP8(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n"    \
                    OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n")   \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
               : "v"(b));
template <int KIND>
__global__ void __launch_bounds__(256) kpk(float* out, int iters, float seed) {
  float2v a0 = {seed, seed}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  float2v b = {seed * 0.5f, seed * 0.25f};
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) { BODYPK("v_pk_add_f32") }
    if (KIND == 1) { BODYPK("v_pk_mul_f32") }
  }
  float2v s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// mixed: v_mov_b32, v_cndmask, v_cvt, v_bfe, dpp
template <int KIND>
__global__ void __launch_bounds__(256) kmisc(float* out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b = seed * 0.5f + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) asm volatile(REP8("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (KIND == 1) asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
    if (KIND == 2) asm volatile(REP8("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_cvt_f32_i32 %4, %4\n v_cvt_f32_i32 %5, %5\n v_cvt_f32_i32 %6, %6\n v_cvt_f32_i32 %7, %7\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (KIND == 3) asm volatile(REP8("v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (KIND == 5) asm volatile("s_mov_b64 s[10:11], 0x5555\n" REP8("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s10", "s11");
    if (KIND == 6) asm volatile(REP8("v_cmp_gt_f32_e64 s[10:11], %0, %8\n v_cmp_gt_f32_e64 s[12:13], %1, %8\n v_cmp_gt_f32_e64 s[14:15], %2, %8\n v_cmp_gt_f32_e64 s[16:17], %3, %8\n v_cmp_gt_f32_e64 s[10:11], %4, %8\n v_cmp_gt_f32_e64 s[12:13], %5, %8\n v_cmp_gt_f32_e64 s[14:15], %6, %8\n v_cmp_gt_f32_e64 s[16:17], %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
    if (KIND == 7) asm volatile(REP8("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (KIND == 8) asm volatile(REP8("v_and_or_b32 %0, %0, %8, 1.0\n v_and_or_b32 %1, %1, %8, 1.0\n v_and_or_b32 %2, %2, %8, 1.0\n v_and_or_b32 %3, %3, %8, 1.0\n v_and_or_b32 %4, %4, %8, 1.0\n v_and_or_b32 %5, %5, %8, 1.0\n v_and_or_b32 %6, %6, %8, 1.0\n v_and_or_b32 %7, %7, %8, 1.0\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (KIND == 9) asm volatile(REP8("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    if (KIND == 4) asm volatile(REP8("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// round-2 additions: what makes an instruction "half rate"?  (a) IEEE mode (v_max_f32 must quiet sNaNs), (b) the 8-byte
// encodings (VOP3 / 32-bit literal), (c) cheap integer VOP2 forms that could replace v_bfe / v_cvt / v_and_or in log2f4.
#define REP8X(A) A A A A A A A A
#define CH8(OP, TAIL)                                                                                                     \
  OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" OP " %4, %4" TAIL "\n" OP   \
     " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
template <int KIND>
__global__ void __launch_bounds__(256) kx(float* out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b = seed * 0.5f + threadIdx.x * 1e-9f;
  if (KIND == 0 || KIND == 7) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");  // IEEE off
  for (int it = 0; it < iters; ++it) {
#define OUTS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)
    if (KIND == 0) asm volatile(REP8X(CH8("v_max_f32", ", %8")) OUTS);                     // v_max_f32, IEEE mode off
    if (KIND == 1) asm volatile(REP8X(CH8("v_add_f32_e64", ", %8")) OUTS);                 // VOP3-encoded add
    if (KIND == 2) asm volatile(REP8X("v_add_f32 %0, 0x3fc00001, %0\n v_add_f32 %1, 0x3fc00001, %1\n v_add_f32 %2, 0x3fc00001, %2\n"
                                      "v_add_f32 %3, 0x3fc00001, %3\n v_add_f32 %4, 0x3fc00001, %4\n v_add_f32 %5, 0x3fc00001, %5\n"
                                      "v_add_f32 %6, 0x3fc00001, %6\n v_add_f32 %7, 0x3fc00001, %7\n") OUTS);  // VOP2 + 32-bit literal
    if (KIND == 3) asm volatile(REP8X(CH8("v_and_b32", ", %8")) OUTS);
    if (KIND == 4) asm volatile(REP8X(CH8("v_lshrrev_b32", ", %8")) OUTS);                 // note: dst = src1 >> src0
    if (KIND == 5) asm volatile(REP8X(CH8("v_or_b32", ", %8")) OUTS);
    if (KIND == 6) asm volatile(REP8X("v_cmp_gt_f32_e32 vcc, %0, %8\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n"
                                      "v_cmp_gt_f32_e32 vcc, %2, %8\n v_addc_co_u32_e32 %3, vcc, %3, %3, vcc\n"
                                      "v_cmp_gt_f32_e32 vcc, %4, %8\n v_addc_co_u32_e32 %5, vcc, %5, %5, vcc\n"
                                      "v_cmp_gt_f32_e32 vcc, %6, %8\n v_addc_co_u32_e32 %7, vcc, %7, %7, vcc\n") OUTS : "vcc");
    if (KIND == 7) asm volatile(REP8X("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n"
                                      "v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n"
                                      "v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n") OUTS);  // IEEE off
    if (KIND == 8) asm volatile(REP8X(CH8("v_sub_f32", ", %8")) OUTS);
    if (KIND == 9) asm volatile(REP8X(CH8("v_fmac_f32", ", %8")) OUTS);                    // VOP2 FMA: dst += src0 * src1
    if (KIND == 10) asm volatile(REP8X(CH8("v_min_f32", ", %8")) OUTS);
    if (KIND == 11) asm volatile(REP8X(CH8("v_max_u32", ", %8")) OUTS);                    // integer max
    if (KIND == 12) asm volatile(REP8X(CH8("v_max_i32", ", %8")) OUTS);
    // compare-bit accumulation without v_cmp: sign of (b - a) shifted into the accumulator by one v_alignbit_b32
    if (KIND == 13) asm volatile(REP8X("v_alignbit_b32 %0, %0, %8, 31\n v_alignbit_b32 %1, %1, %8, 31\n v_alignbit_b32 %2, %2, %8, 31\n"
                                       "v_alignbit_b32 %3, %3, %8, 31\n v_alignbit_b32 %4, %4, %8, 31\n v_alignbit_b32 %5, %5, %8, 31\n"
                                       "v_alignbit_b32 %6, %6, %8, 31\n v_alignbit_b32 %7, %7, %8, 31\n") OUTS);
    if (KIND == 14) asm volatile(REP8X("v_sub_f32 %0, %8, %1\n v_alignbit_b32 %1, %1, %0, 31\n"
                                       "v_sub_f32 %2, %8, %3\n v_alignbit_b32 %3, %3, %2, 31\n"
                                       "v_sub_f32 %4, %8, %5\n v_alignbit_b32 %5, %5, %4, 31\n"
                                       "v_sub_f32 %6, %8, %7\n v_alignbit_b32 %7, %7, %6, 31\n") OUTS);
    if (KIND == 15) asm volatile(REP8X(CH8("v_lshl_or_b32", ", 1, %8")) OUTS);
    if (KIND == 16) asm volatile(REP8X(CH8("v_lshl_add_u32", ", 1, %8")) OUTS);
    if (KIND == 17) asm volatile(REP8X(CH8("v_ashrrev_i32", ", %8")) OUTS);
    if (KIND == 18) asm volatile(REP8X(CH8("v_add_u32", ", %8")) OUTS);
    if (KIND == 19) asm volatile(REP8X(CH8("v_xor_b32", ", %8")) OUTS);
    if (KIND == 20) asm volatile(REP8X(CH8("v_bfi_b32", ", %8, %8")) OUTS);
    if (KIND == 21) asm volatile(REP8X(CH8("v_perm_b32", ", %8, %8")) OUTS);
    if (KIND == 22) asm volatile(REP8X(CH8("v_min3_f32", ", %8, %8")) OUTS);
    if (KIND == 23) asm volatile(REP8X(CH8("v_med3_f32", ", %8, %8")) OUTS);
    if (KIND == 24) asm volatile(REP8X(CH8("v_mul_legacy_f32", ", %8")) OUTS);
    if (KIND == 25) asm volatile(REP8X(CH8("v_subrev_f32", ", %8")) OUTS);
    if (KIND == 26) asm volatile(REP8X(CH8("v_lshlrev_b32", ", %8")) OUTS);
    if (KIND == 27) asm volatile(REP8X(CH8("v_add3_u32", ", %8, %8")) OUTS);
    // round 5: the instruction kinds of the prefilter kernels (hhv_prefilter.hip) - integer clamp / max forms, SDWA and DPP adds,
    // packed 16-bit - and what a MIXED stream of a half-rate and a full-rate kind costs (do they overlap?)
    if (KIND == 28) asm volatile(REP8X(CH8("v_med3_i32", ", 0, %8")) OUTS);
    if (KIND == 29) asm volatile(REP8X(CH8("v_max3_i32", ", %8, %8")) OUTS);
    if (KIND == 30) asm volatile(REP8X("v_add_u32_sdwa %0, sext(%8), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                                       "v_add_u32_sdwa %1, sext(%8), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                                       "v_add_u32_sdwa %2, sext(%8), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                                       "v_add_u32_sdwa %3, sext(%8), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                                       "v_add_u32_sdwa %4, sext(%8), %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                                       "v_add_u32_sdwa %5, sext(%8), %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                                       "v_add_u32_sdwa %6, sext(%8), %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                                       "v_add_u32_sdwa %7, sext(%8), %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n") OUTS);
    if (KIND == 31) asm volatile(REP8X("v_add_u32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                       "v_add_u32_dpp %2, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                       "v_add_u32_dpp %4, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %5, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                       "v_add_u32_dpp %6, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %7, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n") OUTS);
    if (KIND == 32) asm volatile(REP8X(CH8("v_cvt_f32_ubyte1", "")) OUTS);
    if (KIND == 33) asm volatile(REP8X(CH8("v_pk_add_u16", ", %8")) OUTS);
    if (KIND == 34) asm volatile(REP8X(CH8("v_pk_max_i16", ", %8")) OUTS);
    if (KIND == 35) asm volatile(REP8X(CH8("v_pk_sub_u16", ", %8 clamp")) OUTS);
    // mixed streams, 1 : 1 - if the two kinds used separate issue slots the pair would cost what the slower one costs alone
    if (KIND == 36) asm volatile(REP8X("v_med3_i32 %0, %0, 0, %8\n v_add_f32 %1, %1, %8\n v_med3_i32 %2, %2, 0, %8\n v_add_f32 %3, %3, %8\n"
                                       "v_med3_i32 %4, %4, 0, %8\n v_add_f32 %5, %5, %8\n v_med3_i32 %6, %6, 0, %8\n v_add_f32 %7, %7, %8\n") OUTS);
    if (KIND == 37) asm volatile(REP8X("v_max_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                                       "v_max_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n") OUTS);
    if (KIND == 38) asm volatile(REP8X("v_max_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                                       "v_max_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n") OUTS);  // 1 slow : 3 fast
    // one residue of the gapless prefilter kernel (W = 5 cells per lane), the round-4 form and the round-5 form, 8 residues per statement
#define PF_OLD "v_bfe_i32 %7, %8, 0, 8\n v_add_u32_sdwa %4, sext(%8), %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n" \
               "v_add_u32_sdwa %3, sext(%8), %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n" \
               "v_add_u32_dpp %7, %4, %7 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
               "v_add_u32_sdwa %2, sext(%8), %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n" \
               "v_add_u32_sdwa %1, sext(%8), %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" \
               "v_med3_i32 %4, %4, 0, s20\n v_med3_i32 %0, %7, 0, s20\n v_max3_i32 %5, %5, %0, %4\n v_med3_i32 %3, %3, 0, s20\n" \
               "v_med3_i32 %2, %2, 0, s20\n v_med3_i32 %1, %1, 0, s20\n v_max3_i32 %5, %5, %1, %2\n v_max_i32 %5, %5, %3\n"
#define PF_NEW "v_mov_b32 %7, s21\n v_mov_b32_dpp %7, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_lshrrev_b32 %6, 24, %8\n" \
               "v_add_u32_sdwa %3, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n" \
               "v_add_u32_sdwa %2, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n" \
               "v_and_b32 %1, 0xff, %8\n v_add_u32 %5, 0x80, %5\n v_add_u32 %7, %1, %7\n v_add_u32 %4, %8, %3\n v_mov_b32 %1, s22\n v_mov_b32 %0, s23\n" \
               "v_med3_i32 %7, %7, %1, %0\n v_add_u32 %6, %6, %3\n v_max_i32 %5, %5, %7\n v_med3_i32 %3, %3, %1, %0\n v_med3_i32 %2, %2, %1, %0\n" \
               "v_med3_i32 %4, %4, %1, %0\n v_med3_i32 %6, %6, %1, %0\n v_max3_i32 %5, %5, %2, %3\n v_max3_i32 %5, %5, %6, %4\n"
    if (KIND == 41) asm volatile("s_mov_b32 s20, 205\n" REP8X(PF_OLD) OUTS : "s20");
    if (KIND == 42) asm volatile("s_mov_b32 s21, 5\n s_mov_b32 s22, 7\n s_mov_b32 s23, 300\n" REP8X(PF_NEW) OUTS : "s21", "s22", "s23");
    // the new form with its slow-kind instructions spread between the fast ones by hand
#define PF_NEW2 "v_mov_b32 %7, s21\n v_mov_b32_dpp %7, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_lshrrev_b32 %6, 24, %8\n" \
               "v_add_u32_sdwa %3, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n v_and_b32 %1, 0xff, %8\n" \
               "v_add_u32_sdwa %2, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n v_add_u32 %5, 0x80, %5\n" \
               "v_med3_i32 %3, %3, s22, %0\n v_add_u32 %7, %1, %7\n v_med3_i32 %2, %2, s22, %0\n v_add_u32 %4, %8, %3\n" \
               "v_med3_i32 %7, %7, s22, %0\n v_add_u32 %6, %6, %3\n v_med3_i32 %4, %4, s22, %0\n v_mov_b32 %0, s23\n v_med3_i32 %6, %6, s22, %0\n" \
               "v_mov_b32 %1, s22\n v_max3_i32 %5, %5, %2, %3\n v_mov_b32 %1, s22\n v_max3_i32 %5, %5, %6, %4\n v_mov_b32 %1, s22\n v_max_i32 %5, %5, %7\n"
    if (KIND == 43) asm volatile("s_mov_b32 s21, 5\n s_mov_b32 s22, 7\n s_mov_b32 s23, 300\n" REP8X(PF_NEW2) OUTS : "s21", "s22", "s23");
    if (KIND == 39) asm volatile(REP8X(CH8("v_sub_u32", ", %8")) OUTS);
    if (KIND == 40) asm volatile(REP8X(CH8("v_min_u32", ", %8")) OUTS);
#undef OUTS
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename F>
static void run(const char* name, F launch, int width) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out;
  hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = cus * wps;  // 256 threads = 4 waves = one per SIMD
    launch(blocks, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)blocks * 4 * iters * 64.0;  // wave-instructions
    const double clk = prop.clockRate * 1e3;                // Hz
    const double ipc_cu = inst / (ms * 1e-3) / clk / cus;
    printf("%-16s waves/SIMD=%d  %.3f ms  %.2f wave-inst/clk/CU  -> %.2f clk per wave-inst per SIMD  (%.1f T lane-ops/s x%d)\n",
           name, wps, ms, ipc_cu, 4.0 / ipc_cu, inst * 64 * width / (ms * 1e-3) / 1e12, width);
  }
  hipFree(out);
}

// (a > b) against the sign bit of (b - a): equal for every pair of non-NaN floats if -inf - -inf gives a NaN with a clear sign bit
__global__ void ksign(const float* v, int n, unsigned* bad, unsigned* nanbits) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const float a = v[i / n], b = v[i % n];
  float d;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(b), "v"(a));
  const unsigned s = __builtin_bit_cast(unsigned, d) >> 31;
  if (s != (a > b ? 1u : 0u)) atomicAdd(bad, 1u);
  if (d != d) atomicOr(nanbits, __builtin_bit_cast(unsigned, d) | 1u);
}

static void sign_check() {
  const float vals[] = {0.0f, -0.0f, 1.0f, -1.0f, 1.0000001f, 3.4028235e38f, -3.4028235e38f, __builtin_inff(), -__builtin_inff(), 1e-45f,
                        -1e-45f, 1.17549435e-38f, -1.17549435e-38f, 1.17549421e-38f, 123.456f, 123.45601f, -77.25f, -77.250008f, 2e-39f, 3e-39f};
  const int n = sizeof(vals) / sizeof(vals[0]);
  float* dv;
  unsigned *dbad, h[2] = {0, 0};
  hipMalloc(&dv, sizeof(vals));
  hipMalloc(&dbad, 8);
  hipMemcpy(dv, vals, sizeof(vals), hipMemcpyHostToDevice);
  hipMemcpy(dbad, h, 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(ksign, dim3((n * n + 255) / 256), dim3(256), 0, 0, dv, n, dbad, dbad + 1);
  hipMemcpy(h, dbad, 8, hipMemcpyDeviceToHost);
  printf("sign(b - a) == (a > b): %d pairs of %d special values, %u mismatches; NaN results OR-ed bits %08x\n", n * n, n, h[0], h[1]);
}

int main_int();
int main(int argc, char** argv) {
  if (argc > 1) return main_int();
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  printf("device %s  CUs %d  clock %.0f MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1e3);
  run("v_add_f32", [](int b, float* o, int it) { hipLaunchKernelGGL(k<0>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_mul_f32", [](int b, float* o, int it) { hipLaunchKernelGGL(k<1>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_max_f32", [](int b, float* o, int it) { hipLaunchKernelGGL(k<2>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_pk_add_f32", [](int b, float* o, int it) { hipLaunchKernelGGL(kpk<0>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 2);
  run("v_pk_mul_f32", [](int b, float* o, int it) { hipLaunchKernelGGL(kpk<1>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 2);
  run("v_mov_b32", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<0>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_cndmask_b32", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<1>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_cvt_f32_i32", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<2>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_mov_b32_dpp", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<3>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_cndmask_e64_s", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<5>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_cmp_gt_f32_e64", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<6>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_bfe_u32", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<7>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_and_or_b32", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<8>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_fma_f32", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<9>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
  run("v_max3_f32", [](int b, float* o, int it) { hipLaunchKernelGGL(kmisc<4>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1);
#define RUNX(NAME, K) run(NAME, [](int b, float* o, int it) { hipLaunchKernelGGL(kx<K>, dim3(b), dim3(256), 0, 0, o, it, 1.0f); }, 1)
  RUNX("v_max_f32 IEEE=0", 0);
  RUNX("v_max3_f32 IEEE=0", 7);
  RUNX("v_min_f32", 10);
  RUNX("v_max_u32", 11);
  RUNX("v_max_i32", 12);
  RUNX("v_add_f32_e64", 1);
  RUNX("v_add_f32 literal", 2);
  RUNX("v_sub_f32", 8);
  RUNX("v_fmac_f32 (VOP2)", 9);
  RUNX("v_and_b32", 3);
  RUNX("v_lshrrev_b32", 4);
  RUNX("v_or_b32", 5);
  RUNX("v_cmp_e32+v_addc", 6);
  RUNX("v_alignbit_b32", 13);
  RUNX("v_sub_f32+v_alignbit", 14);
  RUNX("v_lshl_or_b32", 15);
  RUNX("v_lshl_add_u32", 16);
  RUNX("v_ashrrev_i32", 17);
  RUNX("v_add_u32", 18);
  RUNX("v_xor_b32", 19);
  RUNX("v_bfi_b32", 20);
  RUNX("v_perm_b32", 21);
  RUNX("v_min3_f32", 22);
  RUNX("v_med3_f32", 23);
  RUNX("v_mul_legacy_f32", 24);
  RUNX("v_subrev_f32", 25);
  RUNX("v_lshlrev_b32", 26);
  RUNX("v_add3_u32", 27);
  sign_check();
  return 0;
}

int main_int() {
  RUNX("v_med3_i32", 28);
  RUNX("v_max3_i32", 29);
  RUNX("v_add_u32_sdwa", 30);
  RUNX("v_add_u32_dpp shr", 31);
  RUNX("v_cvt_f32_ubyte1", 32);
  RUNX("v_pk_add_u16", 33);
  RUNX("v_pk_max_i16", 34);
  RUNX("v_pk_sub_u16 clamp", 35);
  RUNX("v_sub_u32", 39);
  RUNX("v_min_u32", 40);
  RUNX("v_add_u32", 18);
  RUNX("med3 : add_f32 1:1", 36);
  RUNX("max_f32 : add_f32 1:1", 37);
  RUNX("max : add/mul 1:3", 38);
  // (a "wave-inst" of the next three is one of 64 statements' instructions: 8 residues x 14 / 20 / 22 instructions; the table's
  // clk figure is per 1/64 of a statement - multiply by 64 / 8 for clk per RESIDUE)
  RUNX("prefilter residue r4", 41);
  RUNX("prefilter residue r5", 42);
  RUNX("prefilter residue r5 spread", 43);
  return 0;
}
