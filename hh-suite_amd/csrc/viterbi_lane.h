// viterbi_lane.h -- per-lane arithmetic of the systolic Viterbi engine.
//
// Design: one wavefront is a 64-stage systolic array.  Lane g owns the R
// consecutive QUERY rows i = g*R+1 .. g*R+R (profile columns and transitions of those rows live in
// the lane's VGPRs for the whole kernel) and the wave streams template columns through the lanes:
// at step s lane g processes stream record r = s - g.  A record is either a template header
// (column 0: finalize the previous template, initialise the DP boundary) or one template column j.
// The only cross-lane traffic per step is the bottom-row state of lane g-1 (5 floats; the running best only in
// steps with a header), pulled through the LDS crossbar (64-lane arrays) or moved with DPP (short-query arrays);
// everything else is lane local.
//
// The arithmetic restates, operation for operation and in the same association, the reference's
//   Viterbi::AlignWithOutCellOff / AlignWithCellOff  src/hhviterbialgorithm.cpp:144-494
//   Viterbi::ScalarProd20Vec                          src/hhviterbi.h:126-161
//   log2f4 (LOG_POLY_DEGREE 4)                        src/hhutil-inl.h:501-541
// Must be compiled with -ffp-contract=off (the reference build has no FMA).
//
// This header is compiled by hipcc into the product kernel and by g++ into tests/emul (a host-side
// lock-step emulation of one wavefront used to debug the schedule without a GPU; test code only).
#pragma once
#include <float.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HHV_DEV __device__ __forceinline__
#define HHV_MEM __device__ __forceinline__
#define HHV_HD __host__ __device__ __forceinline__
#define HHV_HDMEM __host__ __device__ __forceinline__
#else
#define HHV_DEV static inline
#define HHV_MEM inline
#define HHV_HD static inline
#define HHV_HDMEM inline
#endif

namespace hhv {

// ---- packed column record (28 dwords = 112 B), same layout for query rows and template columns ----
//  [0..19] p[k][a]                          profile column k
//  [20..24] tr[k-1][M2M,M2D,D2M,D2D,I2M]    transitions leaving column k-1
//  [25..26] tr[k][I2I,M2I]                  insert transitions of column k
//  [27]     meta (template stream only)
// This is the operand set of one DP cell: src/hhviterbialgorithm.cpp:182-188 (query side) and
// :222-228 (template side) read exactly slots 2..6 of column k-1 and slots 0..1 of column k.
constexpr int REC_DW = 28;
constexpr int REC_M2M = 20, REC_M2D = 21, REC_D2M = 22, REC_D2D = 23, REC_I2M = 24, REC_I2I = 25, REC_M2I = 26,
              REC_META = 27;
constexpr int32_t META_HDR = (int32_t)0x80000000;  // header record: [0] = template index, [1] = Lt
constexpr int32_t META_LAST = 0x40000000;          // column record of j == Lt
// header record only: global mode, the template's own last column does not take part in the maximisation (it is not the
// last column of its SIMD batch in the reference, src/hhviterbialgorithm.cpp:462-486 - see hhv_set_global_batch).  Carried
// through the template in bit 30 of LaneState::tid (template indices stay below 2^30).
constexpr int32_t META_NOLASTCOL = 0x20000000;
constexpr int32_t TID_NOLASTCOL = 0x40000000;
constexpr int32_t TID_MASK = 0x3FFFFFFF;
constexpr int32_t META_JMASK = 0x0000FFFF;         // j (template lengths are limited to 65535)
// secondary-structure indices of template column j, exactly the per-column bytes HMMSimd::MapHMMVector
// precomputes (src/hhhmmsimd.cpp:132-135): pred_index = ss_pred*MAXCF + ss_conf (0..43), dssp_index = ss_dssp (0..7)
constexpr int META_PRED_SHIFT = 16, META_PRED_MASK = 0x3F;
constexpr int META_DSSP_SHIFT = 22, META_DSSP_MASK = 0x7;

constexpr float NEG_MAX = -FLT_MAX;

HHV_DEV float bits2f(uint32_t u) {
  union { uint32_t u; float f; } x;
  x.u = u;
  return x.f;
}
HHV_DEV uint32_t f2bits(float f) {
  union { uint32_t u; float f; } x;
  x.f = f;
  return x.u;
}
HHV_DEV float fmax2(float a, float b) {
#if defined(__HIPCC__)
  return __builtin_fmaxf(a, b);  // v_max_f32 / v_max3_f32; no NaNs occur on this path
#else
  return a > b ? a : b;
#endif
}

// ---- backtrace entry (8 bytes per lane per column, R <= 5 cells) -----------------------------------------------------
// The kernel does not build the reference's backtrace byte (src/hhviterbimatrix.h:35-48) in the inner loop; it only
// RECORDS nine comparisons of a cell, one bit each, at fixed positions of the entry.  The byte is decoded where it is read
// (trace kernel, hhv_backtrace_matrix) by bt_decode below.
//   phase A1, rows R-1 .. 0, five bits per row (the MM predecessor, see below);
//            rows R-1..1 -> lo[7(R-1)-1 : 2(R-1)], row 0 -> hi[2R+6 : 2R+2]
//   phase A2, rows R-1 .. 0, two bits per row: GD: ga > gb, IM: ia > ib;  rows R-1..1 -> lo[2(R-1)-1 : 0], row 0 -> hi[2R+1 : 2R]
//   phase C, rows 0 .. R-1, two bits per row: DG: da > db, MI: ma > mb                 -> hi[2R-1:0]
// The MM predecessor.  The reference keeps a running maximum over smin, c1 .. c5 and the code of the LAST candidate that was
// strictly greater (src/hhviterbialgorithm.cpp:241-273, MAX2 with the byte codes :17-20) - which is the FIRST candidate that
// reaches the final maximum m.  Two encodings of that:
//   BT_MM_FIRST_EQUAL  e0 = (m > smin), e_k = (c_k == m) for k = 1..4: m comes from two v_max3 + one v_max (the five
//                      maxima of the running form are three instructions), the flags from v_cmp + v_addc pairs
//   BT_MM_RUNNING      c1 > smin, c2 > m1, .. c5 > m4 with the v_cmpx form of bt_max below
// and two forms of a pairwise maximum with its bit (GD, IM, DG, MI, and the running MM form):
//   BT_PAIR_ADDC       v_max + v_cmp (VCC) + v_addc (VCC as carry-in: acc = 2 acc + bit)
//   BT_PAIR_CMPX       v_cmpx narrows EXEC to the lanes with a > m, a masked v_mov takes a there, a masked v_or sets the bit,
//                      EXEC comes back from an SGPR pair: one VOPC + two plain VOP2 instead of three VOPC / carry instructions
// Which is faster depends on the kernel around it (profiles/r3_ab.txt, one session each): the 64-lane arrays run FIRST_EQUAL
// + ADDC (-3 %: 20.4 -> 19.7 ms per 100 k templates; the v_cmpx forms gain nothing there), the short-query arrays RUNNING +
// CMPX (Lq 150: 12.0 -> 11.0 ms, -8 %; FIRST_EQUAL + CMPX 11.2, FIRST_EQUAL + ADDC 11.7).  The macros are the A/B switches.
//   BT_MM_FIRST_EQUAL_NEG / BT_PAIR_SIGN (round 4, 64-lane variants without cell-off)  the same nine facts taken from the SIGN
//                      of a difference, shifted in by v_alignbit_b32 (acc = {acc, d} >> 31): a > b is the sign bit of b - a,
//                      and c_k == m (c_k <= m) is the INVERTED sign bit of c_k - m, stored inverted (bt_decode flips the four
//                      bits).  No VOPC, no VCC, no carry chain.  Exact because no value compared here is ever -0 (b - a = -0
//                      for a = +0, b = -0 would claim a > b): the DP boundaries are formed as j * (0 - egt), i.e. +0 where the
//                      reference has -0 (equal values, and x + t is -0 only for x = t = -0: by induction no state is), and
//                      two -inf operands (a NaN difference) need two masked cells: the cell-off variants keep v_cmp.
enum { BT_MM_RUNNING = 1, BT_MM_FIRST_EQUAL = 3, BT_MM_FIRST_EQUAL_NEG = 5, BT_PAIR_ADDC = 0, BT_PAIR_CMPX = 1, BT_PAIR_SIGN = 2 };
#ifndef HHV_BT_SIGN
#define HHV_BT_SIGN 1
#endif
#ifndef HHV_BT_MM64
#define HHV_BT_MM64 BT_MM_FIRST_EQUAL
#endif
#ifndef HHV_BT_PAIR64
#define HHV_BT_PAIR64 BT_PAIR_ADDC
#endif
#ifndef HHV_BT_MMS
#define HHV_BT_MMS BT_MM_RUNNING
#endif
#ifndef HHV_BT_PAIRS
#define HHV_BT_PAIRS BT_PAIR_CMPX
#endif
// The encoding a kernel variant writes; the host records it with the backtrace buffer (hhv_tset::bt_mm) and hands it to the
// kernels that decode.  (The local five-row cell-off + secondary-structure variant of the short-query arrays has no register
// left for the running form's maximum: it flags.)
HHV_HD constexpr bool bt_sign_mode(int W, bool celloff) { return HHV_BT_SIGN && W == 64 && !celloff && HHV_BT_MM64 == BT_MM_FIRST_EQUAL; }
HHV_HD constexpr int bt_mm_mode(int W, int R, bool local, bool celloff, bool ss) {
  return bt_sign_mode(W, celloff) ? (int)BT_MM_FIRST_EQUAL_NEG
         : W == 64 ? HHV_BT_MM64 : (R == 5 && local && celloff && ss) ? (int)BT_MM_FIRST_EQUAL : HHV_BT_MMS;
}
HHV_HD constexpr int bt_pair_mode(int W, bool celloff) { return bt_sign_mode(W, celloff) ? (int)BT_PAIR_SIGN : W == 64 ? HHV_BT_PAIR64 : HHV_BT_PAIRS; }

// acc = 2 acc + (a > b) / + (a == b)
HHV_DEV void bt_push(uint32_t& acc, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_cmp_gt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
  acc = (acc << 1) | (a > b ? 1u : 0u);
#endif
}
HHV_DEV void bt_push_eq(uint32_t& acc, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("v_cmp_eq_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
#else
  acc = (acc << 1) | (a == b ? 1u : 0u);
#endif
}
// acc = 2 acc + signbit(x - y): for x, y that are never -0 (and not both -inf) this is acc = 2 acc + (y > x)
HHV_DEV void bt_push_sign(uint32_t& acc, float x, float y) {
  const float d = x - y;
#if defined(__HIP_DEVICE_COMPILE__)
  acc = __builtin_amdgcn_alignbit(acc, f2bits(d), 31);
#else
  acc = (acc << 1) | (f2bits(d) >> 31);
#endif
}
// (Compares into SGPR pairs of their own - v_cmp_e64 - with the v_addc_e64 that consume them issued as a block behind, so that
// no instruction waits for VCC: built and measured, 19.3 -> 26.4 ms per 100 k templates; a VALU write to an SGPR pair is far more
// expensive than the VCC round trip it avoids.  profiles/r3_ab.txt, ab-r3-4.)
// if (a > m) { m = a; acc |= bit; }   (bit: a compile-time constant after unrolling; exec_save: EXEC of the caller's block)
HHV_DEV void bt_max(uint32_t& acc, float& m, float a, uint32_t bit, uint64_t exec_save) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_cmpx_gt_f32_e32 vcc, %2, %0\n\tv_mov_b32_e32 %0, %2\n\tv_or_b32_e32 %1, %3, %1\n\ts_mov_b64 exec, %4"
               : "+v"(m), "+v"(acc) : "v"(a), "i"(bit), "s"(exec_save) : "vcc");
#else
  (void)exec_save;
  if (a > m) {
    m = a;
    acc |= bit;
  }
#endif
}
HHV_DEV uint64_t bt_exec() {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_read_exec();
#else
  return 0;
#endif
}
// entry -> the reference's byte for row r of the lane (bits 0-2 MM predecessor, 8 GD, 16 IM, 32 DG, 64 MI)
HHV_HD uint32_t bt_decode(uint64_t entry, int r, int R, int mm_mode) {
  const uint32_t lo = (uint32_t)entry, hi = (uint32_t)(entry >> 32);
  const uint32_t f7 = r >= 1 ? (((lo >> (2 * (R - 1) + 5 * (r - 1))) & 0x1Fu) << 2) | ((lo >> (2 * (r - 1))) & 3u)
                             : (hi >> (2 * R)) & 0x7Fu;
  const uint32_t c2 = (hi >> (2 * (R - 1 - r))) & 3u;
  uint32_t b = 0;
  if (mm_mode == BT_MM_FIRST_EQUAL_NEG) {
    // e0 = (m > smin) as is; bits 5..2 hold c_k < m, i.e. NOT (c_k == m)
    const uint32_t g7 = f7 ^ 0x3Cu;
    if (g7 & 0x40u) b = (g7 & 0x20u) ? 2 : (g7 & 0x10u) ? 3 : (g7 & 0x08u) ? 4 : (g7 & 0x04u) ? 5 : 6;
  } else if (mm_mode == BT_MM_FIRST_EQUAL) {
    // e0 = (m > smin); the first of c1..c4 that equals m, c5 if none does: codes 2 (MM), 3 (GD), 4 (IM), 5 (DG), 6 (MI)
    if (f7 & 0x40u) b = (f7 & 0x20u) ? 2 : (f7 & 0x10u) ? 3 : (f7 & 0x08u) ? 4 : (f7 & 0x04u) ? 5 : 6;
  } else {
    if (f7 & 0x40u) b = 2;  // c1 > smin            : MM
    if (f7 & 0x20u) b = 3;  // c2 > max so far      : GD
    if (f7 & 0x10u) b = 4;  //                        IM
    if (f7 & 0x08u) b = 5;  //                        DG
    if (f7 & 0x04u) b = 6;  //                        MI
  }
  b |= (f7 & 0x02u) ? 8u : 0u;
  b |= (f7 & 0x01u) ? 16u : 0u;
  b |= (c2 & 0x02u) ? 32u : 0u;
  b |= (c2 & 0x01u) ? 64u : 0u;
  return b;
}

// The constants of log2f4.  On the device the kernel pins them in SGPRs (hhv_stream_kernel.h: Log2Consts::pinned()): as
// 32-bit literals they make 8-byte instructions, and a literal-carrying VOP2 costs a wave ~0.3 clk more than one with an SGPR
// operand (tools/gen_replay_ubench.py: the emission phase replayed with and without its literals, 4.81 vs 4.53 clk per
// instruction; 30 such instructions per step).  Same values either way.
struct Log2Consts {
  float c4, c3, c2, c1;  // -0.10725..., 0.68824..., -1.75647..., 2.61761...  (src/hhutil-inl.h:530-535)
  float ebias;           // 8388608 + 127
  uint32_t expor;        // bits of 8388608.0f >> 9 (the high operand of v_alignbit_b32 .., 23)
  uint32_t mant;         // mantissa mask
  HHV_HDMEM static Log2Consts literal() {
    Log2Consts k;
    k.c4 = -0.107254423828329604454f;
    k.c3 = 0.688243882994381274313f;
    k.c2 = -1.75647175389045657003f;
    k.c1 = 2.61761038894603480148f;
    k.ebias = 8388735.0f;
    k.expor = 0x4B000000u >> 9;
    k.mant = 0x007FFFFFu;
    return k;
  }
};

#if defined(HHV_EMISSION_FMA)
// OPT-IN BUILD (make lib_fma -> libhhviterbi_hip_fma.so), never the default and never what parity or bench.py's `value` are
// judged on: the emission score with fused multiply-adds - the 16 accumulating products of the 20-term sum and log2f4's
// polynomial - 20 VALU instructions per cell less.  One rounding per term instead of two: Viterbi scores move by ~1e-5 - and, rarely,
// by 3.9e-4, beyond BASELINE.json's 1e-4: log2f4 jumps by 3.93e-4 at every power of two, and a product whose last bit changes across
// one takes the jump (1.2e7 templates: max 3.98e-4, 12 end points changed; profiles/r6_fast_mode_bound.json, tools/fast_mode_bound.py).  The
// oracle restates this arithmetic too (hho_set_emission_mode(2)), so the build is still tested bit for bit - against that.
HHV_DEV float fmadd(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#endif
// src/hhutil-inl.h:509-541, one rounding per operation
HHV_DEV float log2f4(float x, const Log2Consts& K) {
  const uint32_t i = f2bits(x);
#if defined(__HIP_DEVICE_COMPILE__)
  // e = float(biased exponent - 127) without v_cvt and without a field extract: ONE v_alignbit_b32 drops the exponent field
  // into the mantissa of 2^23 - alignbit(K, bits, 23) = (bits >> 23) | (K << 9) with K = 0x4B000000 >> 9, i.e. the float
  // 8388608 + E exactly - and 8388608 + 127 is subtracted; every step is exact, so e is the float the reference's
  // int -> float conversion gives.  (bits >> 23 is the exponent field because x >= 0: x is a sum of products of profile
  // values, which are probabilities / odds - hhv_upload_templates and hhv_set_query refuse negative ones.  Round 2 masked the
  // field with v_bfe_u32 + v_or_b32: one VALU instruction more per cell, 1 % of the kernel.)
  const float e = bits2f(__builtin_amdgcn_alignbit(K.expor, i, 23)) - K.ebias;
#else
  const float e = (float)((int32_t)((i & 0x7F800000u) >> 23) - 127);
#endif
  const float m = bits2f((i & K.mant) | 0x3F800000u);
#if defined(HHV_EMISSION_FMA)
  // opt-in build only (see dot20 below): Horner steps and the final p * (m - 1) + e as fused multiply-adds
  float p = fmadd(K.c4, m, K.c3);
  p = fmadd(p, m, K.c2);
  p = fmadd(p, m, K.c1);
  return fmadd(p, m - 1.0f, e);
#else
  float p = K.c4 * m;
  p = p + K.c3;
  p = p * m;
  p = p + K.c2;
  p = p * m;
  p = p + K.c1;
  p = p * (m - 1.0f);
  return p + e;
#endif
}
HHV_DEV float log2f4(float x) { return log2f4(x, Log2Consts::literal()); }

// src/hhviterbi.h:126-161: four partial accumulators, (r0+r1)+(r2+r3)
HHV_DEV float dot20(const float* q, const float* t) {
  float r0 = t[0] * q[0];
  float r1 = t[1] * q[1];
  float r2 = t[2] * q[2];
  float r3 = t[3] * q[3];
#pragma unroll
  for (int k = 4; k < 20; k += 4) {
#if defined(HHV_EMISSION_FMA)
    r0 = fmadd(t[k + 0], q[k + 0], r0);
    r1 = fmadd(t[k + 1], q[k + 1], r1);
    r2 = fmadd(t[k + 2], q[k + 2], r2);
    r3 = fmadd(t[k + 3], q[k + 3], r3);
#else
    r0 = t[k + 0] * q[k + 0] + r0;
    r1 = t[k + 1] * q[k + 1] + r1;
    r2 = t[k + 2] * q[k + 2] + r2;
    r3 = t[k + 3] * q[k + 3] + r3;
#endif
  }
  r0 = r0 + r1;
  r2 = r2 + r3;
  return r0 + r2;
}

struct Params {
  float negq, negt, shift;  // 0 - egq, 0 - egt (formed once by the caller: Params::set_gaps), par.shift
  int Lq;
  Log2Consts lg = Log2Consts::literal();
  // The DP boundaries -j * egt and -i * egq (src/hhviterbialgorithm.cpp:144-153,161-173) are formed as j * negt and i * negq
  // with negt = 0 - egt: the same value bit for bit - except that a zero penalty gives +0 where the reference's int -> float
  // -> multiply gives -0 (equal as values; no comparison or sum of the recurrence can tell them apart).  With that, NO state
  // of the DP is ever -0 (x + t is -0 only for x = t = -0), which the sign-bit form of the backtrace flags relies on.
  HHV_HDMEM void set_gaps(float egq, float egt) {
    negq = 0.0f - egq;
    negt = 0.0f - egt;
  }
};

// query rows owned by one lane (row i = i0 + r)
template <int R>
struct QRows {
  float p[R][20];
  float m2m[R], m2d[R], d2m[R], d2d[R], i2m[R];  // of row i-1
  float i2i[R], m2i[R];                          // of row i
  HHV_MEM void load(const float* rows /* R consecutive 28-dword records */) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* w = rows + r * REC_DW;
#pragma unroll
      for (int a = 0; a < 20; ++a) p[r][a] = w[a];
      m2m[r] = w[REC_M2M];
      m2d[r] = w[REC_M2D];
      d2m[r] = w[REC_D2M];
      d2d[r] = w[REC_D2D];
      i2m[r] = w[REC_I2M];
      i2i[r] = w[REC_I2I];
      m2i[r] = w[REC_M2I];
    }
  }
};

// what lane g reads from lane g-1 at the top of every step (lane 0: the DP boundary row 0)
struct Incoming {
  float MM, GD, IM, DG, MI;  // state of row i0-1 at the column lane g-1 has just finished
  float fs;                  // finalized best of lanes < g (valid on header steps)
  int fpos;                  // (i2 << 16) | j2
};

// Row-0 sums that read the diagonal (i0-1, j-1), i.e. what lane g-1 handed over ONE step ago (st.dGD / dIM / dDG).  They
// are taken at the top of a step, before this step's hand-off: the kernel then lets the DPP move write the new hand-off
// values straight into st.dGD / dIM / dDG (no second register set, no copy at the end of the step; the first lane of an
// array is never written by the move and keeps its boundary value -FLT_MAX for the whole kernel).
struct DiagSums {
  float t2, x3, x4;  // dGD + q.m2m[0], dIM + q.i2m[0], dDG + q.d2m[0]   (src/hhviterbialgorithm.cpp:241-273, first add)
};

template <int R>
struct LaneState {
  float MM[R], GD[R], IM[R], DG[R], MI[R];  // own rows, column j-1 (the previous step)
  float dMM, dGD, dIM, dDG, dMI;            // row i0-1, column j-1 (top-row diagonal)
  // SHARE variants: (MM(i-1,j-1) + q.M2M(i-1)) and (MI(i-1,j-1) + q.M2M(i-1)).  The same two sums feed MI(i,j-1)
  // one step earlier (:358-366) and MM(i,j) now (:241-273), so they are computed once and carried.
  float aMM[R], aMI[R];
  float bs[R];                              // running best per row ...
  int bj[R];                                // ... and its column (local mode: every row; global: row Lq only, slot 0)
  float fs;                                 // finalized best over lanes <= g for the template just finished
  int fpos;
  int tid;   // template index of the template being processed (-1 before the first header)
  int jlast; // column index of the last processed column record
  HHV_MEM void reset() {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      MM[r] = GD[r] = IM[r] = DG[r] = MI[r] = NEG_MAX;
      aMM[r] = aMI[r] = NEG_MAX;
      bs[r] = NEG_MAX;
      bj[r] = 0;
    }
    dMM = dGD = dIM = dDG = dMI = NEG_MAX;
    fs = NEG_MAX;
    fpos = 0;
    tid = -1;
    jlast = 0;
  }
};

template <int R>
HHV_DEV DiagSums lane_diag(const LaneState<R>& st, const QRows<R>& q) {
  DiagSums d;
  d.t2 = st.dGD + q.m2m[0];
  d.x3 = st.dIM + q.i2m[0];
  d.x4 = st.dDG + q.d2m[0];
  return d;
}

// boundary row 0 as seen by lane 0 (src/hhviterbialgorithm.cpp:144-153,161): MM(0,j) = -j*egt, the rest
// -FLT_MAX.  On a header step the value becomes the diagonal of cell (1,1), which the reference
// initialises as -(i-1)*egq = -0*egq (:161).
HHV_DEV Incoming boundary_incoming(int32_t meta, int j /* meta & META_JMASK */, const Params& P) {
  Incoming in;
  if (meta < 0) in.MM = 0.0f;  // -(0) * egq: +0 (0 * negq would be -0 for a positive penalty)
  else in.MM = (float)(j) * P.negt;
  in.GD = in.IM = in.DG = in.MI = NEG_MAX;
  in.fs = NEG_MAX;
  in.fpos = 0;
  return in;
}
HHV_DEV Incoming boundary_incoming(int32_t meta, const Params& P) { return boundary_incoming(meta, meta & META_JMASK, P); }

struct TemplateResult {
  float score;
  int i2, j2;
  int tid;
};

// Header record: finish the previous template (combine this lane's best with the prefix best that
// flows down the lanes; the lane owning row Lq emits the result) and set the column-0 boundary
// (:161-173: MM(i,0) = -i*egq, other states -FLT_MAX).  Returns true if `res` must be written.
template <int R, bool LOCAL, bool SHARE>
HHV_DEV bool lane_header(LaneState<R>& st, const QRows<R>& q, const Incoming& in, int i0, int new_tid, const Params& P,
                         bool is_last_lane, TemplateResult& res) {
  bool emit = false;
  if (st.tid >= 0) {
    // best over own rows in row order, strict '>' (ties keep the smaller row, then the smaller column:
    // the reference's row-major scan with strict '>' at :423-455,462-486)
    float s = NEG_MAX;
    int pos = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = i0 + r;
      float cs;
      int cj;
      if (LOCAL) {
        cs = st.bs[r];
        cj = st.bj[r];
      } else {
        // global: rows < Lq contribute their last-column cell (still held in MM[r]); row Lq its best column
        if (i == P.Lq) {
          cs = st.bs[0];
          cj = st.bj[0];
        } else {
          cs = (st.tid & TID_NOLASTCOL) ? NEG_MAX : st.MM[r];  // NEG_MAX never wins the strict '>' below
          cj = st.jlast;
        }
      }
      const bool take = (i <= P.Lq) && (cs > s);
      s = take ? cs : s;
      pos = take ? ((i << 16) | cj) : pos;
    }
    // rows of lanes < g come first in scan order: keep theirs unless strictly beaten
    const bool own = s > in.fs;
    st.fs = own ? s : in.fs;
    st.fpos = own ? pos : in.fpos;
    if (is_last_lane) {
      res.score = st.fs;
      res.i2 = st.fpos >> 16;
      res.j2 = st.fpos & 0xFFFF;
      res.tid = st.tid & TID_MASK;
      emit = true;
    }
  }
  st.tid = new_tid;
  st.jlast = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    st.MM[r] = (float)(i0 + r) * P.negq;
    st.GD[r] = st.IM[r] = st.DG[r] = st.MI[r] = NEG_MAX;
    st.bs[r] = NEG_MAX;
    st.bj[r] = 0;
  }
  st.dMM = in.MM;
  st.dGD = in.GD;
  st.dIM = in.IM;
  st.dDG = in.DG;
  st.dMI = in.MI;
  if (SHARE) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      st.aMM[r] = (r ? st.MM[r - 1] : in.MM) + q.m2m[r];
      st.aMI[r] = (r ? st.MI[r - 1] : in.MI) + q.m2m[r];
    }
  }
  return emit;
}

// Operand source of a column step: where the template record and (QL sources) the lane's "second tier" query
// transitions come from.  ArraySrc reads plain arrays (host emulation, tests); the kernel's LdsColumn
// (hhv_kernels.hip) reads the LDS ring with inline-asm ds_read_b128 and puts the waits where the values are used.
//   tr(k)        k = 0..6: the record's M2M, M2D, D2M, D2D, I2M, I2I, M2I      (needed first, phase A)
//   get_p(tp)    the 20 profile values, valid from phase B on (a source may still be waiting for them before)
//   qa(r, w)     QL only: w = 0 m2i, 1 i2i of row r (phase A);   qc(r, w): w = 0 m2d, 1 d2d (phase C)
//   begin_column / before_A2 / before_B / before_C   issue and wait points of an asynchronous source
struct ArraySrc {
  static constexpr bool QL = false;
  const float* rec;
  HHV_MEM void begin_column() {}
  HHV_MEM float tr(int k) const { return rec[REC_M2M + k]; }
  HHV_MEM void before_A2() {}
  HHV_MEM void before_B() {}
  HHV_MEM void get_p(float* tp) const {
    for (int a = 0; a < 20; ++a) tp[a] = rec[a];
  }
  HHV_MEM void before_C() {}
  HHV_MEM float qa(int, int) const { return 0.0f; }
  HHV_MEM float qc(int, int) const { return 0.0f; }
  // the hand-off of the step as phase C and the end of the column see it (a source may deliver it late: LdsColumn)
  template <class State>
  HHV_MEM Incoming resolve(const Incoming& in, State&) const { return in; }
};

// Source of the secondary-structure values of a column (SS variants): PtrSs = R values the caller has gathered already (host
// emulation, short-query arrays); an ASYNC source (hhv_stream_kernel.h SsLds) is asked for them behind the profile's wait and
// delivers in front of phase C.
struct PtrSs {
  static constexpr bool ASYNC = false;
  const float* v;
  HHV_MEM void issue() {}
  HHV_MEM void wait() {}
  HHV_MEM float get(int r) const { return v[r]; }
};

// One template column j for the R rows of this lane.
//   src      : the 28-dword column record (and, for QL sources, four of the lane's query transitions per row)
//   cellbits : CELLOFF only - byte r bit 7 set = cell (i0+r, j) excluded (same byte matrix the
//              backtrace is written to, src/hhviterbialgorithm.cpp:373-392)
//   returns  : BT only - the 9 compare bits of each of the R cells (layout and decoding: bt_push / bt_decode above)
//
// The R cells are evaluated in three phases so that every state register can be updated in place
// (no loop-carried copies): A (rows bottom-up) everything that reads only column j-1 state - the five
// MM candidates, GD and IM; B the R emission scores (independent dot products = the ILP of the
// kernel); C (rows top-down) MM += S, then DG and MI which chain through the row above.
//   ssx      : SS only - ssx.get(r) = ssw * S[q_ss(i0+r)][t_ss(j)], the secondary-structure term of the ...AndSS
//              builds (src/hhviterbialgorithm.cpp:194-213,278-280), added as ss + log2f4(..) like the reference
//   BTM / BTP : BT only - encoding of the MM predecessor and form of the pairwise maxima (bt_mm_mode / bt_pair_mode of the array width)
template <int R, bool LOCAL, bool BT, bool CELLOFF, bool SHARE, bool SS, int BTM, int BTP, class Src, class Ssx>
HHV_DEV uint64_t lane_column(LaneState<R>& st, const QRows<R>& q, const Incoming& in, const DiagSums& ds, Src& src, int j,
                             int i0, int r_last /* (Lq-1) % R */, const Params& P, uint64_t cellbits, Ssx& ssx) {
  constexpr bool QL = Src::QL;
  const float smin = LOCAL ? 0.0f : NEG_MAX;
  src.begin_column();
  const float tM2M = src.tr(0), tM2D = src.tr(1), tD2M = src.tr(2), tD2D = src.tr(3), tI2M = src.tr(4), tI2I = src.tr(5),
              tM2I = src.tr(6);
  float cmax[R];
  uint32_t acc_lo = 0, acc_hi = 0;  // BT: compare bits of rows R-1..1 / of row 0 and phase C (layout: bt_decode)
  // ---- phase A, rows R-1 .. 0: reads (i-1, j-1) = old state of the row above and (i, j-1) = own old state
  if (BT) {
    // backtrace variants, two sweeps: A1 = the MM candidates of all rows (they read only registers and the record head),
    // A2 = the GD / IM updates, which overwrite what A1 read and need the query's {m2i, i2i} - with a QL source those
    // come from LDS and have had the whole of A1 to arrive (src.before_A2()).
    const uint64_t ex = bt_exec();
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      const float dMM = r ? st.MM[r - 1] : st.dMM, dMI = r ? st.MI[r - 1] : st.dMI;
      // :241-273 (row 0: the first add of c2 / c3 / c4 was taken before the hand-off, lane_diag)
      const float c1 = (SHARE ? st.aMM[r] : (dMM + q.m2m[r])) + tM2M;
      const float c2 = (r ? st.GD[r - 1] + q.m2m[r] : ds.t2) + tD2M;
      const float c3 = (r ? st.IM[r - 1] + q.i2m[r] : ds.x3) + tM2M;
      const float c4 = (r ? st.DG[r - 1] + q.d2m[r] : ds.x4) + tM2M;
      const float c5 = (SHARE ? st.aMI[r] : (dMI + q.m2m[r])) + tI2M;
      uint32_t& acc = r ? acc_lo : acc_hi;
      if (BTM == BT_MM_FIRST_EQUAL_NEG) {
        const float mm = fmax2(fmax2(fmax2(fmax2(fmax2(smin, c1), c2), c3), c4), c5);  // v_max3, v_max3, v_max
        bt_push_sign(acc, smin, mm);  // m > smin
        bt_push_sign(acc, c1, mm);    // c_k < m (the equality flag, inverted)
        bt_push_sign(acc, c2, mm);
        bt_push_sign(acc, c3, mm);
        bt_push_sign(acc, c4, mm);
        cmax[r] = mm;
      } else if (BTM == BT_MM_FIRST_EQUAL) {
        const float mm = fmax2(fmax2(fmax2(fmax2(fmax2(smin, c1), c2), c3), c4), c5);  // v_max3, v_max3, v_max
        bt_push(acc, mm, smin);
        bt_push_eq(acc, c1, mm);
        bt_push_eq(acc, c2, mm);
        bt_push_eq(acc, c3, mm);
        bt_push_eq(acc, c4, mm);
        cmax[r] = mm;
      } else {
        // (the same five positions through fixed masks; the accumulators of this mode are only ever or-ed into)
        const int top = r ? 7 * (R - 1) - 1 - 5 * (R - 1 - r) : 6 + 2 * R;  // position of the row's first bit
        float mm = smin;
        bt_max(acc, mm, c1, 1u << (top - 0), ex);
        bt_max(acc, mm, c2, 1u << (top - 1), ex);
        bt_max(acc, mm, c3, 1u << (top - 2), ex);
        bt_max(acc, mm, c4, 1u << (top - 3), ex);
        bt_max(acc, mm, c5, 1u << (top - 4), ex);
        cmax[r] = mm;
      }
    }
    static_assert(!BT || BTM == BT_MM_FIRST_EQUAL || BTM == BT_MM_FIRST_EQUAL_NEG || BTP == BT_PAIR_CMPX, "the running form sets its bits in place");
    static_assert((BTM == BT_MM_FIRST_EQUAL_NEG) == (BTP == BT_PAIR_SIGN) || !BT, "the sign forms come together");
    if (BTM == BT_MM_FIRST_EQUAL && BTP == BT_PAIR_CMPX) {
      // the flags were shifted in, the pairwise bits are set in place: make room for them
      acc_lo <<= 2 * (R - 1);
      acc_hi <<= 2 + 2 * R;
    }
    src.before_A2();
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      // :307-332 (GD and IM read only the cell to the left)
      const float lMM = st.MM[r];
      const float ga = lMM + tM2D;
      float gb = st.GD[r] + tD2D;
      const float qm2i = QL ? src.qa(r, 0) : q.m2i[r], qi2i = QL ? src.qa(r, 1) : q.i2i[r];
      const float ia = (lMM + qm2i) + tM2M;
      float ib = (st.IM[r] + qi2i) + tM2M;
      uint32_t& acc = r ? acc_lo : acc_hi;
      if (BTP == BT_PAIR_CMPX) {
        const int top = r ? 2 * (R - 1) - 1 - 2 * (R - 1 - r) : 1 + 2 * R;
        bt_max(acc, gb, ga, 1u << top, ex);
        bt_max(acc, ib, ia, 1u << (top - 1), ex);
        st.GD[r] = gb;
        st.IM[r] = ib;
      } else if (BTP == BT_PAIR_SIGN) {
        bt_push_sign(acc, gb, ga);  // ga > gb
        bt_push_sign(acc, ib, ia);
        st.GD[r] = fmax2(ga, gb);
        st.IM[r] = fmax2(ia, ib);
      } else {
        bt_push(acc, ga, gb);
        bt_push(acc, ia, ib);
        st.GD[r] = fmax2(ga, gb);
        st.IM[r] = fmax2(ia, ib);
      }
    }
  } else {
#pragma unroll
    for (int r = R - 1; r >= 0; --r) {
      const float dMM = r ? st.MM[r - 1] : st.dMM, dMI = r ? st.MI[r - 1] : st.dMI;
      // :241-273.  Score only: c1, c3 and c4 add the same tM2M.  x -> fl(x + t) is monotonic (round to nearest), hence
      // max(fl(x1+t), fl(x3+t), fl(x4+t)) == fl(max(x1, x3, x4) + t) bit for bit: one addition instead of three.
      // (The backtrace variants cannot do this: their compare bits must see the rounded sums, which can tie where the x do not.)
      const float x1 = SHARE ? st.aMM[r] : (dMM + q.m2m[r]), x3 = r ? st.IM[r - 1] + q.i2m[r] : ds.x3,
                  x4 = r ? st.DG[r - 1] + q.d2m[r] : ds.x4;
      const float c2 = (r ? st.GD[r - 1] + q.m2m[r] : ds.t2) + tD2M;
      const float c5 = (SHARE ? st.aMI[r] : (dMI + q.m2m[r])) + tI2M;
      cmax[r] = fmax2(smin, fmax2(fmax2(fmax2(fmax2(x1, x3), x4) + tM2M, c2), c5));
      // :307-332 (GD and IM read only the cell to the left); IM with the same monotonicity argument
      const float lMM = st.MM[r];
      const float ga = lMM + tM2D, gb = st.GD[r] + tD2D;
      const float qm2i = QL ? src.qa(r, 0) : q.m2i[r], qi2i = QL ? src.qa(r, 1) : q.i2i[r];
      st.IM[r] = fmax2(lMM + qm2i, st.IM[r] + qi2i) + tM2M;
      st.GD[r] = fmax2(ga, gb);
    }
  }
  // ---- phase B: :277-283
  src.before_B();
  float tp[20];
  src.get_p(tp);
  float S[R];
  constexpr bool SSA = SS && Ssx::ASYNC;  // the table values arrive in front of phase C: the two additions behind log2f4 wait for them
  if (SSA) ssx.issue();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float v = log2f4(dot20(q.p[r], tp), P.lg);
    if (SS && !SSA) v = ssx.get(r) + v;
    S[r] = SSA ? v : v + P.shift;
  }
  // ---- phase C, rows 0 .. R-1: (i-1, j) = new state of the row above
#if defined(HHV_EXP_TIMING) && defined(__HIP_DEVICE_COMPILE__)
  src.stamp_B(S[0], S[R - 1]);
#endif
  src.before_C();
  if (SSA) {
    ssx.wait();
#pragma unroll
    for (int r = 0; r < R; ++r) S[r] = (ssx.get(r) + S[r]) + P.shift;  // (ss + log2f4(..)) + shift, :278-283
  }
  const Incoming up = src.resolve(in, st);  // MM / DG / MI of row i0-1 in this column; GD / IM / DG for the next column's diagonal
  float uMM = up.MM, uDG = up.DG, uMI = up.MI;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float mm = cmax[r] + S[r];
    // :340-366
    const float qm2d = QL ? src.qc(r, 0) : q.m2d[r], qd2d = QL ? src.qc(r, 1) : q.d2d[r];
    const float da = uMM + qm2d, db = uDG + qd2d;
    const bool cmpx = BT && BTP == BT_PAIR_CMPX;
    float dg = cmpx ? db : fmax2(da, db);
    const float sa = uMM + q.m2m[r], sb = uMI + q.m2m[r];
    const float ma = sa + tM2I, mb = sb + tI2I;
    float mi = cmpx ? mb : fmax2(ma, mb);
    if (SHARE) {
      st.aMM[r] = sa;
      st.aMI[r] = sb;
    }
    if (cmpx) {
      const uint64_t exc = bt_exec();
      bt_max(acc_hi, dg, da, 1u << (2 * (R - 1 - r) + 1), exc);
      bt_max(acc_hi, mi, ma, 1u << (2 * (R - 1 - r)), exc);
    } else if (BT && BTP == BT_PAIR_SIGN) {
      bt_push_sign(acc_hi, db, da);  // da > db
      bt_push_sign(acc_hi, mb, ma);
    } else if (BT) {
      bt_push(acc_hi, da, db);
      bt_push(acc_hi, ma, mb);
    }
    if (CELLOFF) {  // :373-392: the masked build adds -FLT_MAX or +0.0f to all five states of every cell
      const float add = ((cellbits >> (8 * r)) & 0x80u) ? NEG_MAX : 0.0f;
      mm = mm + add;
      st.GD[r] = st.GD[r] + add;
      st.IM[r] = st.IM[r] + add;
      dg = dg + add;
      mi = mi + add;
    }
    if (LOCAL) {  // :423-455, strict '>' keeps the earliest column of a row
      const bool up = mm > st.bs[r];
      st.bs[r] = up ? mm : st.bs[r];
      st.bj[r] = up ? j : st.bj[r];
    }
    st.MM[r] = mm;
    st.DG[r] = dg;
    st.MI[r] = mi;
    uMM = mm;
    uDG = dg;
    uMI = mi;
  }
  if (!LOCAL) {
    // global alignment: row Lq is maximised over all columns (:192,423); r_last is wave uniform
    float v = st.MM[0];
#pragma unroll
    for (int r = 1; r < R; ++r) v = (r == r_last) ? st.MM[r] : v;
    const bool up = v > st.bs[0];
    st.bs[0] = up ? v : st.bs[0];
    st.bj[0] = up ? j : st.bj[0];
  }
  st.jlast = j;
  st.dMM = up.MM;
  st.dGD = up.GD;
  st.dIM = up.IM;
  st.dDG = up.DG;
  st.dMI = up.MI;
  return BT ? ((uint64_t)acc_hi << 32) | acc_lo : 0;
}

}  // namespace hhv
