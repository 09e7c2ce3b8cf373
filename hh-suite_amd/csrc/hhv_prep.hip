// hhv_prep.hip -- on-device PrepareTemplateHMM: raw template HMMs (as HMM::Read leaves
// them) -> prepared, packed record stream, once per query, so that a search never touches template data on
// the host.  Restates, operation for operation (fp32/fp64 mix included), the reference's
//   HMM::AddTransitionPseudocounts    src/hhhmm.cpp:1722-1806   (fpow2 src/util-inl.h:190-215, fast_log2 :108-130)
//   HMM::PreparePseudocounts          src/hhhmm.cpp:1811-1815   (ScalarProd20, plain branch, src/hhhit-inl.h:125-131)
//   HMM::AddAminoAcidPseudocounts     src/hhhmm.cpp:1874-1964   (pcm 0 - 3; pcm 2 with pcc != 1: tau comes from the host)
//   HMM::CalculateAminoAcidBackground src/hhhmm.cpp:1854-1868   (NormalizeTo1 src/util-inl.h:277-291)
//   HMM::IncludeNullModelInHMM        src/hhhmm.cpp:2059-2144   (columnscore 0..3)
// in the order of PrepareTemplateHMM (src/hhfunc.cpp:165-202, HHM format).
//
// Fused kernel (templates up to 1300 columns): one workgroup per template - (1) one lane per raw column
// (elementwise: transitions, g = R*f, p = (1-tau) f + tau g) into LDS, (2) lanes 0..19 of the first wave accumulate
// pav[a] over the columns in the reference's order, (3) all lanes divide by the null model and emit the packed
// 28-dword records (+ header).  Longer templates take the same three steps as two kernels with the intermediate in HBM.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

#include "hhv_internal.h"
#include "viterbi_lane.h"

namespace hhv {

// src/util-inl.h:190-215
__device__ __forceinline__ float fpow2_dev(float x) {
  if (x >= FLT_MAX_EXP) return FLT_MAX;
  if (x <= FLT_MIN_EXP) return 0.0f;
  const float tx = (x - 0.5f) + (float)(3 << 22);
  const int lx = (int)(f2bits(tx) - 0x4b400000u);
  const float dx = x - (float)lx;
  float y = dx * 0.0134929f;
  y = 0.0520749f + y;
  y = dx * y;
  y = 0.241404f + y;
  y = dx * y;
  y = 0.693019f + y;
  y = dx * y;
  y = 1.0f + y;
  return bits2f(f2bits(y) + ((uint32_t)lx << 23));
}

__device__ __forceinline__ float fast_log2_p(float x, const float* __restrict__ lg2, const float* __restrict__ diff) {
  if (x <= 0) return -100000;
  const uint32_t u = f2bits(x);
  const int aa = (int)((u & 0x7F800000u) >> 23) - 0x7f;
  const int b = (int)((u & 0x007FE000u) >> 13);
  const int c = (int)(u & 0x00001FFFu);
  return ((float)aa + lg2[b]) + diff[b] * (float)c;
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// R[a][b] in the pair layout prep_column reads (two rows of R side by side)
__device__ __forceinline__ void stage_R_pairs(const float* __restrict__ R, float* __restrict__ sR, int tid, int nthreads) {
  for (int q = tid; q < 400; q += nthreads) {
    const int aa = q / 20, b = q - aa * 20;
    sR[((aa >> 1) * 20 + b) * 2 + (aa & 1)] = R[q];
  }
}

// reference enum order of tr[][7], src/hhdecl.h:68
enum { T_M2M = 0, T_M2I = 1, T_M2D = 2, T_I2M = 3, T_I2I = 4, T_D2M = 5, T_D2D = 6 };

// One raw column -> its prepared transitions T[7] and pseudocount-mixed profile P[20] (columns >= 1).
// lg2 / diff: fast_log2's tables (the fused kernel keeps a copy in LDS: seven look-ups with a different index in every lane per
// column - as global loads they kept the CU's address unit busy for a third of the kernel's time)
__device__ __forceinline__ void prep_column(const PrepArgs& a, const float (&raw)[RAW_DW], int64_t raw_col, const float* __restrict__ sR,
                                            const float* __restrict__ lg2, const float* __restrict__ diff,
                                            float* __restrict__ P, float* __restrict__ T) {
  const int i = __builtin_bit_cast(int32_t, raw[RAW_J]) & META_JMASK;
  const int L = __builtin_bit_cast(int32_t, raw[RAW_L]);

  // ---- AddTransitionPseudocounts (:1743-1785)
  float t[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) t[k] = raw[RAW_TR + k];
  if (a.gapb > 0) {
    float pM2D, pM2I;
    pM2D = pM2I = (float)((double)a.gapd * 0.0286);
    const float pM2M = 1 - pM2D - pM2I;
    const float pI2I = (float)(1.0 * (double)a.gape / ((double)(a.gape - 1) + 1.0 / 0.75));
    const float pI2M = 1 - pI2I;
    const float pD2D = pI2I;
    const float pD2M = 1 - pD2D;
    const float nM = raw[RAW_NEFF + 0], nI = raw[RAW_NEFF + 1], nD = raw[RAW_NEFF + 2];
    float p0 = (nM - 1) * fpow2_dev(t[T_M2M]) + a.gapb * pM2M;
    float p1 = (nM - 1) * fpow2_dev(t[T_M2D]) + a.gapb * pM2D;
    float p2 = (nM - 1) * fpow2_dev(t[T_M2I]) + a.gapb * pM2I;
    if (i == 0 || i == L) p1 = p2 = 0;
    float sum = p0 + p1 + p2 + FLT_MIN;
    t[T_M2M] = fast_log2_p(p0 / sum, lg2, diff);
    t[T_M2D] = fast_log2_p(p1 / sum, lg2, diff) * a.gapf;
    t[T_M2I] = fast_log2_p(p2 / sum, lg2, diff) * a.gapg;
    p0 = nI * fpow2_dev(t[T_I2M]) + a.gapb * pI2M;
    p1 = nI * fpow2_dev(t[T_I2I]) + a.gapb * pI2I;
    sum = p0 + p1 + FLT_MIN;
    t[T_I2M] = fast_log2_p(p0 / sum, lg2, diff);
    t[T_I2I] = fast_log2_p(p1 / sum, lg2, diff) * a.gapi;
    p0 = nD * fpow2_dev(t[T_D2M]) + a.gapb * pD2M;
    p1 = nD * fpow2_dev(t[T_D2D]) + a.gapb * pD2D;
    if (i == L) p1 = 0;
    sum = p0 + p1 + FLT_MIN;
    t[T_D2M] = fast_log2_p(p0 / sum, lg2, diff);
    t[T_D2D] = fast_log2_p(p1 / sum, lg2, diff) * a.gaph;
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) T[k] = t[k];

  // ---- PreparePseudocounts + AddAminoAcidPseudocounts (columns 1..L)
  if (i >= 1) {
    float f[20];
#pragma unroll
    for (int b = 0; b < 20; ++b) f[b] = raw[RAW_F + b];
    float tau = 0.0f;
    if (a.pcm == 1) tau = a.pca;
    if (a.pcm == 2)  // :1898-1909; pcc != 1 needs libm's powf: the host has evaluated tau for every raw column
      tau = a.tau ? a.tau[raw_col] : (float)fmin(1.0, (double)a.pca / (1. + (double)(raw[RAW_NEFF + 0] / a.pcb)));
    if (a.pcm == 3) {  // :1911-1919, constant-diversity pseudocounts: float arithmetic, the maximum with the double 0.0
      const float x = raw[RAW_NEFF + 0] / a.pcb;
      tau = (float)fmax(0.0, (double)(a.pca * ((1.0f - x) + (a.pcc * x) * (1.0f - x))));
    }
    if (a.pcm == 0) {
#pragma unroll
      for (int aa = 0; aa < 20; ++aa) P[aa] = f[aa];
    } else {
      // g[a] = ScalarProd20(R[a], f[i]): tj[0]*qi[0] + tj[1]*qi[1] + ... left to right - for TWO rows a, a + 1 of R at a time with
      // the packed fp32 multiply and add of gfx950 (v_pk_mul_f32 / v_pk_add_f32: two independent IEEE operations per lane and
      // instruction, each rounded as the scalar one - no fused multiply-add): 400 instructions a column instead of 800.
      // sR holds R as pairs: sR[(a / 2 * 20 + b) * 2 + (a & 1)] = R[a][b]
      const f32x2* R2 = reinterpret_cast<const f32x2*>(sR);
#pragma unroll
      for (int ap = 0; ap < 10; ++ap) {
        const f32x2* Ra = R2 + ap * 20;
        // (the pairs are read from LDS here, ten 16-byte broadcast reads per two rows: left to itself the compiler keeps all of R in
        // accumulation registers across the columns of a wave and pays a v_accvgpr_read per value and use)
        asm volatile("" ::: "memory");
        f32x2 g = f32x2{f[0], f[0]} * Ra[0];
#pragma unroll
        for (int b = 1; b < 20; ++b) g = g + f32x2{f[b], f[b]} * Ra[b];
        P[2 * ap] = (float)((1. - (double)tau) * (double)f[2 * ap] + (double)(tau * g.x));
        P[2 * ap + 1] = (float)((1. - (double)tau) * (double)f[2 * ap + 1] + (double)(tau * g.y));
      }
    }
  }
}

// one raw column (128 bytes, 128-byte aligned) as eight 16-byte loads
__device__ __forceinline__ void load_raw_column(const PrepArgs& a, int64_t col, float (&v)[RAW_DW]) {
  const float4* src = reinterpret_cast<const float4*>(a.raw + (size_t)col * RAW_DW);
#pragma unroll
  for (int q = 0; q < RAW_DW / 4; ++q) {
    const float4 x = src[q];
    v[4 * q + 0] = x.x;
    v[4 * q + 1] = x.y;
    v[4 * q + 2] = x.z;
    v[4 * q + 3] = x.w;
  }
}

// slot k of the output set is raw template src(k), whose columns start at rin(k) in the raw block
__device__ __forceinline__ int prep_src(const PrepArgs& a, int k) { return a.src ? a.src[k] : k; }
__device__ __forceinline__ int64_t prep_rin(const PrepArgs& a, int k) { return a.raw_off[prep_src(a, k)]; }

// CalculateAminoAcidBackground + IncludeNullModelInHMM + emission of the packed stream for template k, by one wavefront
// (`lane`); p(j, a) = Pbuf[j * PS + a], tr(j, slot) = Tbuf[j * 8 + slot] (LDS in the fused kernel, global in the split one)
template <int PS>
__device__ __forceinline__ void finalize_template(const PrepArgs& a, int k, int lane, const float* Pbuf, const float* Tbuf,
                                                  float* s_pav, float* s_pnul) {
  const int L = a.L[k];
  // ---- CalculateAminoAcidBackground (:1854-1868): 20 independent sequential sums over the columns
  if (lane < 20) {
    float pav = a.pb[lane] * 100.0f / a.neff_hmm[prep_src(a, k)];
    const float* col = Pbuf + (size_t)PS + lane;
    int i = 1;
    // the additions are strictly sequential (fp32 order of the reference); the loads are batched ahead of them
    for (; i + 7 <= L; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = col[(size_t)u * PS];
#pragma unroll
      for (int u = 0; u < 8; ++u) pav += v[u];
      col += 8 * PS;
    }
    for (; i <= L; ++i) {
      pav += *col;
      col += PS;
    }
    s_pav[lane] = pav;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    float sum = 0.0f;
    for (int q = 0; q < 20; ++q) sum += s_pav[q];
    if (sum != 0.0f) {
      const float fac = (float)(1.0 / (double)sum);
      for (int q = 0; q < 20; ++q) s_pav[q] *= fac;
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- IncludeNullModelInHMM (:2069-2093)
  if (lane < 20) {
    float pn;
    switch (a.columnscore) {
      case 1: pn = (float)(0.5 * (double)(a.q_pav[lane] + s_pav[lane])); break;
      case 2: pn = s_pav[lane]; break;
      case 3: pn = a.q_pav[lane]; break;
      default: pn = a.pb[lane]; break;
    }
    s_pnul[lane] = pn;
    if (a.pav_out) a.pav_out[(size_t)k * 20 + lane] = s_pav[lane];
  }
}

// emission of the packed stream: header + L column records (layout: viterbi_lane.h); `tid` of `nthreads` (multiple of 32)
template <int PS>
__device__ __forceinline__ void emit_records(const PrepArgs& a, int k, int tid, int nthreads, const float* Pbuf,
                                             const float* Tbuf, const float* s_pnul) {
  const int64_t c0 = a.rec_off[k], rin = prep_rin(a, k);
  const int L = a.L[k];
  float* hdr = a.records + (size_t)c0 * REC_DW;
  if (tid < REC_DW) {
    float v = 0.0f;
    if (tid == 0) v = __builtin_bit_cast(float, (int32_t)k);
    if (tid == 1) v = __builtin_bit_cast(float, (int32_t)L);
    if (tid == REC_META) v = __builtin_bit_cast(float, META_HDR);
    hdr[tid] = v;
  }
  // one record per 32 threads per iteration: threads 0..27 of each group of 32 write the 28 dwords of column j
  const int w = tid & 31;
  if (w < REC_DW) {
    for (int j = 1 + (tid >> 5); j <= L; j += nthreads >> 5) {
      float v;
      if (w < 20) {
        v = Pbuf[(size_t)j * PS + w] / s_pnul[w];
      } else if (w < REC_META) {
        // [20..24] tr[j-1][M2M,M2D,D2M,D2D,I2M], [25..26] tr[j][I2I,M2I]
        const int src_col = (w <= REC_I2M) ? j - 1 : j;
        const int slot = (w == REC_M2M) ? T_M2M : (w == REC_M2D) ? T_M2D : (w == REC_D2M) ? T_D2M
                       : (w == REC_D2D) ? T_D2D : (w == REC_I2M) ? T_I2M : (w == REC_I2I) ? T_I2I : T_M2I;
        v = Tbuf[(size_t)src_col * 8 + slot];
      } else {
        int32_t meta = j | (__builtin_bit_cast(int32_t, a.raw[(size_t)(rin + j) * RAW_DW + RAW_SS]) & 0x01FF0000);
        if (j == L) meta |= META_LAST;
        v = __builtin_bit_cast(float, meta);
      }
      a.records[(size_t)(c0 + j) * REC_DW + w] = v;
    }
  }
}

// ---- split path (templates too long for the LDS of the fused kernel): P1 over the columns of the listed templates ...
__global__ void __launch_bounds__(256) hhv_prep_columns_kernel(PrepArgs a) {
  // R[20][20] is read 400 times per column with wave-uniform indices: stage it in LDS (broadcast reads)
  __shared__ __attribute__((aligned(16))) float sR[400];
  stage_R_pairs(a.R, sR, threadIdx.x, 256);
  __syncthreads();
  const int k = a.ids[blockIdx.x];
  const int64_t rin = prep_rin(a, k);  // the intermediate is indexed like the raw block
  const int L = a.L[k];
  for (int i = threadIdx.x; i <= L; i += 256) {
    float rawv[RAW_DW];
    load_raw_column(a, rin + i, rawv);
    prep_column(a, rawv, rin + i, sR, a.lg2, a.diff, a.p_tmp + (size_t)(rin + i) * 20, a.tr_tmp + (size_t)(rin + i) * 8);
  }
}

// ... and P2, one wavefront per template
__global__ void __launch_bounds__(64) hhv_prep_finalize_kernel(PrepArgs a) {
  __shared__ float s_pav[20];
  __shared__ float s_pnul[20];
  const int k = a.ids[blockIdx.x];
  const int64_t rin = prep_rin(a, k);
  finalize_template<20>(a, k, threadIdx.x, a.p_tmp + (size_t)rin * 20, a.tr_tmp + (size_t)rin * 8, s_pav, s_pnul);
  __syncthreads();
  emit_records<20>(a, k, threadIdx.x, 64, a.p_tmp + (size_t)rin * 20, a.tr_tmp + (size_t)rin * 8, s_pnul);
}

// ---- fused path: one workgroup of 256 threads per template; the mixed profile p[L+1][20] (row stride 21 dwords:
// conflict-free for one-dword-per-lane accesses) and the prepared transitions stay in LDS between the three steps, so
// the only HBM traffic is the raw column (128 B) in and the packed record (112 B) out.
// Round 6: the three steps are spread over the four wavefronts BY COST.  Until then wave 0 ran two of the five 64-column chunks
// of step 1 (300 columns), all of step 2 and its share of step 3 - twice the instructions of the other waves - and the wave 0 of
// every resident workgroup sat on the same SIMD (4.36 ms per 100 000 templates = 0.25 of the VALU issue rate,
// profiles/r6a_prep_summary.txt).  Now: chunk c of step 1 to wave c mod 4; step 2 (the 20 sequential sums) to the wave with the
// least work so far; the chunks of step 3 greedily to the least loaded wave.  Step 3 itself works one LANE per column (20
// divisions, the record assembled in registers, seven 16-byte stores) instead of one lane per dword of the record (three
// divergent branches per record, 32 lanes for 28 dwords); the raw column comes in as eight 16-byte loads, and its secondary-
// structure bits wait in the unused eighth slot of the column's transitions instead of being fetched again.
constexpr int PREP_PS = 21;
constexpr int PREP_TAB = 1028;
// NT templates per workgroup of 4 * NT wavefronts (NT = 2: the 8 KB of fast_log2 tables and R are shared by two templates, and
// four templates stay resident per CU - with one template per workgroup the tables cost a quarter of the occupancy)
// Who does what (static: a plan worked out per workgroup cost more scalar instructions than the kernel has vector ones), NW = 4 NT
// wavefronts:
//   step 1  work item q (the chunks of the workgroup's templates, in order) -> wave q mod NW: the low waves get the extra items
//   step 2  template t -> wave NW - 1 - t (the high waves)
//   step 3  work item q -> the waves that had no step 2, round-robin
// (Measured, 100 000 templates of 300 columns, profiles/r6_prep_summary.txt: 4.36 ms before; lane-per-column emission + 16-byte loads
// and stores + steps spread by cost 2.78; + tables in LDS, one template per workgroup, three workgroups per CU 2.95; two templates
// per workgroup 2.60; ten wavefronts - one chunk each - 2.74: the balance inside a workgroup is not what limits it any more.)
// (Later in round 6: R f as packed pairs 2.52 ms.  The timing build below - every wave's clocks per step - shows a workgroup spending
// 8 % staging the tables, 45 % in step 1, 24 % in step 2 with six of its eight waves idle at the barrier - 300 dependent additions at
// ~100 clocks each, because the SIMD the lone wave sits on is shared with three busy waves of the other workgroup -, 23 % in step 3.
// A rewrite with persistent workgroups (tables staged once), the raw columns of a wave's next chunk requested a chunk ahead, the
// profile transposed in LDS and step 2 reading 16 bytes at a time was bit-exact and SLOWER, 3.02 ms: the 32 registers of the
// prefetch push the kernel past the 128 of four waves per SIMD (36 spilled), and step 2 stayed at ~100 clocks a column - it is issue
// arbitration, not LDS latency.  profiles/r6_prep_steps.txt.)
// (Last session of round 6, three more probes, A/B in one call each.  (a) The templates' ids / offsets and the raw column of a wave's
// first chunk requested BEFORE the tables are staged, so that the HBM latency runs under the staging - bit-exact, 2.74 ms against
// 2.51: slower, reproducibly.  (b) The tables' seven loads per thread sent at once instead of load - wait - write rounds: 2.50 = 2.51.
// (c) Step 2 cut to 16 columns (wrong sums, timing only): 2.46 against 2.59 in that call - the lone chain wave that a quarter of the
// timing build's clocks point at is worth 5 %, the other resident workgroup fills the CU meanwhile.  What the kernel costs is its
// instruction mix: ~1 490 vector instructions per column and lane, of them 780 packed fp32 at ~4.4 clocks and ~250 fp64 / division
// steps - about 1.1 ms of issue at 16 waves per CU - plus 0.9-1.15 ms of HBM time that the two resident workgroups overlap only
// partly (2.5 ms ~ the sum).  NOTES_r6.md section 10.)
#ifdef HHV_PREP_TIMING
// measurement build (make lib_variant NAME=pt FLAGS=-DHHV_PREP_TIMING): clocks every wavefront spends in the steps of the fused kernel
// and at its barriers, summed per wave index: [wave][0 tables, 1 step 1, 2 barrier, 3 step 2, 4 barrier, 5 step 3], [wave][7] = count
__device__ unsigned long long g_prep_clk[16 * 8];
#define PREP_T(slot)                                                                      \
  {                                                                                       \
    const unsigned long long t_now = __builtin_readcyclecounter();                        \
    if (lane == 0) atomicAdd(&g_prep_clk[wave * 8 + (slot)], t_now - t_last);             \
    t_last = t_now;                                                                       \
  }
#else
#define PREP_T(slot)
#endif
template <int NT>
__global__ void __launch_bounds__(1024) hhv_prep_fused_kernel(PrepArgs a, int n_ids) {
  const int NTH = blockDim.x, NW = NTH >> 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sR = reinterpret_cast<float*>(smem);  // [400]
  float* s_pav0 = sR + 400;                    // [NT][20]
  float* s_pnul0 = s_pav0 + 20 * NT;           // [NT][20] (+ padding to a multiple of 16 bytes)
  float* s_lg2 = s_pnul0 + 20 * NT + 4;        // [1028] fast_log2's tables
  float* s_diff = s_lg2 + PREP_TAB;            // [1028]
  float* sT0 = s_diff + PREP_TAB;              // [NT][(maxL+1)][8]: tr[7] + the column's meta bits
  float* sP0 = sT0 + (size_t)NT * a.lds_cols * 8;  // [NT][(maxL+1)][21]
#ifdef HHV_PREP_TIMING
  unsigned long long t_last = __builtin_readcyclecounter();
#endif
  stage_R_pairs(a.R, sR, threadIdx.x, NTH);
  for (int q = threadIdx.x; q < 1025; q += NTH) s_lg2[q] = a.lg2[q], s_diff[q] = a.diff[q];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  PREP_T(0)
  int k[NT], L[NT], n_chunks[NT];
  int64_t rin[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int slot = blockIdx.x * NT + t;
    k[t] = slot < n_ids ? a.ids[slot] : -1;
    L[t] = k[t] >= 0 ? a.L[k[t]] : 0;
    rin[t] = k[t] >= 0 ? prep_rin(a, k[t]) : 0;
    n_chunks[t] = k[t] >= 0 ? (L[t] + 1 + 63) >> 6 : 0;
  }
  const int W3_N = (NW > NT && !a.step3_all) ? NW - NT : NW;  // step 3 on the waves that had no step 2, or on all
  // ---- step 1: one lane per raw column
  {
    int q = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float* sT = sT0 + (size_t)t * a.lds_cols * 8;
      float* sP = sP0 + (size_t)t * a.lds_cols * PREP_PS;
      for (int c = 0; c < n_chunks[t]; ++c, ++q) {
        if ((q % NW) != wave) continue;
        const int i = (c << 6) + lane;
        if (i <= L[t]) {
          float rawv[RAW_DW];
          load_raw_column(a, rin[t] + i, rawv);
          prep_column(a, rawv, rin[t] + i, sR, s_lg2, s_diff, sP + (size_t)i * PREP_PS, sT + (size_t)i * 8);
          sT[(size_t)i * 8 + 7] = rawv[RAW_SS];
        }
      }
    }
  }
  PREP_T(1)
  __syncthreads();
  PREP_T(2)
  // ---- step 2: the column sums, in the reference's order
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (k[t] >= 0 && wave == NW - 1 - t) {
      // the whole workgroup waits for this one chain of dependent additions: highest issue priority on its SIMD, which it shares
      // with waves of the other resident workgroup (HHV_PREP_NO_PRIO in the environment of the build: without)
#ifndef HHV_PREP_NO_PRIO
      __builtin_amdgcn_s_setprio(3);
#endif
      finalize_template<PREP_PS>(a, k[t], lane, sP0 + (size_t)t * a.lds_cols * PREP_PS, sT0 + (size_t)t * a.lds_cols * 8, s_pav0 + 20 * t, s_pnul0 + 20 * t);
#ifndef HHV_PREP_NO_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  PREP_T(3)
  __syncthreads();
  PREP_T(4)
  // ---- step 3: header + one record per column
  int q3 = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (k[t] < 0) continue;
    const float* sT = sT0 + (size_t)t * a.lds_cols * 8;
    const float* sP = sP0 + (size_t)t * a.lds_cols * PREP_PS;
    const float* s_pnul = s_pnul0 + 20 * t;
    const int64_t c0 = a.rec_off[k[t]];
    if (threadIdx.x < REC_DW) {
      float v = 0.0f;
      if (threadIdx.x == 0) v = __builtin_bit_cast(float, (int32_t)k[t]);
      if (threadIdx.x == 1) v = __builtin_bit_cast(float, (int32_t)L[t]);
      if (threadIdx.x == REC_META) v = __builtin_bit_cast(float, META_HDR);
      a.records[(size_t)c0 * REC_DW + threadIdx.x] = v;
    }
    for (int c = 0; c < n_chunks[t]; ++c, ++q3) {
      if ((q3 % W3_N) != wave) continue;
      const int j = (c << 6) + lane;  // (column 0 has no record of its own: the header above)
      if (j < 1 || j > L[t]) continue;
      float rec[REC_DW];
      const float* pj = sP + (size_t)j * PREP_PS;
#pragma unroll
      for (int q = 0; q < 20; ++q) rec[q] = pj[q] / s_pnul[q];
      // [20..24] tr[j-1][M2M,M2D,D2M,D2D,I2M], [25..26] tr[j][I2I,M2I]
      const float4* tq = reinterpret_cast<const float4*>(sT + (size_t)(j - 1) * 8);
      const float4 m0 = tq[0], m1 = tq[1], n0 = tq[2], n1 = tq[3];  // tr[j-1][0..7], tr[j][0..7]
      rec[REC_M2M] = m0.x;   // T_M2M = 0
      rec[REC_M2D] = m0.z;   // T_M2D = 2
      rec[REC_D2M] = m1.y;   // T_D2M = 5
      rec[REC_D2D] = m1.z;   // T_D2D = 6
      rec[REC_I2M] = m0.w;   // T_I2M = 3
      rec[REC_I2I] = n1.x;   // T_I2I = 4
      rec[REC_M2I] = n0.y;   // T_M2I = 1
      int32_t meta = j | (__builtin_bit_cast(int32_t, n1.w) & 0x01FF0000);
      if (j == L[t]) meta |= META_LAST;
      rec[REC_META] = __builtin_bit_cast(float, meta);
      float4* dst = reinterpret_cast<float4*>(a.records + (size_t)(c0 + j) * REC_DW);
#pragma unroll
      for (int q = 0; q < REC_DW / 4; ++q) dst[q] = make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]);
    }
  }
  PREP_T(5)
#ifdef HHV_PREP_TIMING
  if (lane == 0) atomicAdd(&g_prep_clk[wave * 8 + 7], 1ull);
#endif
}

// (nt templates per workgroup)
size_t prepare_fused_lds(int max_L, int nt) { return (size_t)(400 + 40 * nt + 4 + 2 * PREP_TAB + (size_t)nt * (max_L + 1) * (8 + PREP_PS)) * sizeof(float); }
size_t prepare_fused_lds(int max_L) { return prepare_fused_lds(max_L, 1); }

// Neff_M of every raw column, compacted (for the host's tau table when pcc != 1, hhv_api_prep.cpp ensure_tau)
__global__ void __launch_bounds__(256) hhv_gather_neff_kernel(const float* __restrict__ raw, int64_t n_cols, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cols; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = raw[i * RAW_DW + RAW_NEFF];
}
int launch_gather_neff(const float* raw, int64_t n_cols, float* out, void* stream) {
  const int blocks = (int)std::min<int64_t>(4096, (n_cols + 255) / 256);
  hipLaunchKernelGGL(hhv_gather_neff_kernel, dim3(std::max(1, blocks)), dim3(256), 0, (hipStream_t)stream, raw, n_cols, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// ids / n_ids of the three length classes (fused with a small LDS footprint, fused with a large one, split)
int launch_prepare(const PrepArgs& a0, const int32_t* const ids[3], const int32_t n_ids[3], const int32_t max_L[3], void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  for (int cls = 0; cls < 2; ++cls) {
    if (n_ids[cls] == 0) continue;
    PrepArgs a = a0;
    a.ids = ids[cls];
    a.lds_cols = max_L[cls] + 1;
    // two templates per workgroup where two such workgroups fit a CU's LDS (four templates resident, as with one template and no
    // tables), else one
    const size_t lds2 = prepare_fused_lds(max_L[cls], 2), lds1 = prepare_fused_lds(max_L[cls], 1);
    // measurement aids: wavefronts per template (4; 5 and 6 - one chunk of a 300-column template per wave - measured no faster:
    // 2.51 / 2.60 / 2.46 ms), step 3 on every wave
    static const int wpt_env = [] { const char* e = getenv("HHV_PREP_WAVES"); return e ? atoi(e) : 0; }();
    static const int w3_env = [] { const char* e = getenv("HHV_PREP_W3ALL"); return e ? atoi(e) : -1; }();
    const int wpt = wpt_env > 0 ? std::min(8, wpt_env) : 4;
    a.step3_all = w3_env >= 0 ? w3_env : 0;
    if (lds2 <= 80 * 1024) {
      (void)hipFuncSetAttribute((const void*)hhv_prep_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
      hipLaunchKernelGGL(hhv_prep_fused_kernel<2>, dim3((n_ids[cls] + 1) / 2), dim3(2 * wpt * 64), lds2, stream, a, n_ids[cls]);
    } else {
      (void)hipFuncSetAttribute((const void*)hhv_prep_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      hipLaunchKernelGGL(hhv_prep_fused_kernel<1>, dim3(n_ids[cls]), dim3(wpt * 64), lds1, stream, a, n_ids[cls]);
    }
  }
  if (n_ids[2]) {
    PrepArgs a = a0;
    a.ids = ids[2];
    hipLaunchKernelGGL(hhv_prep_columns_kernel, dim3(n_ids[2]), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(hhv_prep_finalize_kernel, dim3(n_ids[2]), dim3(LANES), 0, stream, a);
  }
#ifdef HHV_PREP_TIMING
  {
    (void)hipStreamSynchronize(stream);
    unsigned long long h[16 * 8], z[16 * 8] = {};
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prep_clk), sizeof(h));
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_prep_clk), z, sizeof(z));
    fprintf(stderr, "hhv_prep_fused_kernel: mean clocks per wave: tables, step 1, barrier, step 2, barrier, step 3\n");
    for (int w = 0; w < 16; ++w)
      if (h[w * 8 + 7])
        fprintf(stderr, "  wave %2d: %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f   (%llu workgroups)\n", w, (double)h[w * 8] / h[w * 8 + 7],
                (double)h[w * 8 + 1] / h[w * 8 + 7], (double)h[w * 8 + 2] / h[w * 8 + 7], (double)h[w * 8 + 3] / h[w * 8 + 7],
                (double)h[w * 8 + 4] / h[w * 8 + 7], (double)h[w * 8 + 5] / h[w * 8 + 7], h[w * 8 + 7]);
  }
#endif
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv
