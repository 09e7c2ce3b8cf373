// hhv_mac.hip -- MAC realignment on MI355X: replaces, for a batch of hits of one query,
//   PosteriorDecoder::forwardAlgorithm   src/hhforwardalgorithm.cpp:10-182   scaled sum-product forward, F_MM kept as float
//   PosteriorDecoder::backwardAlgorithm  src/hhbackwardalgorithm.cpp:10-135  backward + posterior = F*B/Pforward
//   PosteriorDecoder::macAlgorithm       src/hhmacalgorithm.cpp:18-179       maximum-accuracy DP on the posteriors
//   PosteriorDecoder::backtraceMAC       src/hhbacktracemac.cpp:113-240      path, per-step score and posterior
//
// One workgroup per hit: one wavefront in the kernels right below (row state in global memory: the longest templates), seven or
// eight in the dataflow kernels further down (row state in LDS).  The reference's arithmetic is double precision with a per-row rescaling
// (scale[i+1] = 1 / (1 + max_j F_MM(i,j))), so rows are processed one after the other; inside a row
//   * everything that depends only on the previous row (MM, DG, MI forward; pmatch, DG, MI backward) is computed by
//     64 lanes for 64 columns at a time, with exactly the reference's expression trees (no FMA contraction);
//   * the two first-order recurrences ALONG the row (GD and IM) and the running total of Pforward are evaluated in the
//     reference's sequential order by a 64-step DPP sweep: in step s every lane recomputes y = a + y(lane-1)*b from its
//     left neighbour (wave_shr:1, lane 0 takes the carry of the previous strip); after step s lanes 0..s are final, and
//     recomputing a final value from final inputs reproduces it, so no masking is needed.  Rounding is therefore the
//     reference's, operation for operation - the kernels are bit-exact against it, not just within tolerance.
// Row state (5 doubles per column, two rows) lives in LDS - in global memory for templates beyond 2046 columns (GROWS);
// F_MM / posteriors (float) and the MAC backtrace codes are matrices in HBM, one per hit.  The hits of a call are launched
// by template-length class (launch_mac below), every class with the LDS footprint of its own longest template and on a
// stream of its own.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdlib.h>
#include <stdio.h>

#include <float.h>

#include "hhv_internal.h"

namespace hhv {

namespace {

enum { T_M2M = 0, T_M2I = 1, T_M2D = 2, T_I2M = 3, T_I2I = 4, T_D2M = 5, T_D2D = 6 };  // src/hhdecl.h:68
enum { MAC_STOP = 0, MAC_MM = 2, MAC_IM = 4, MAC_MI = 6 };                              // ViterbiMatrix codes
// strips of 64 columns whose HBM operands are fetched a whole row ahead in the STAGE variants (templates up to 832 columns;
// the template itself fits into LDS up to ~800)
constexpr int MAC_PRE = 13;
// PosteriorDecoder::m_back_forward_matrix_threshold (src/hhposteriordecoder.cpp:62): a float
constexpr float MAC_LIST_THRESHOLD = 0.0001f;

__device__ __forceinline__ double shr1_d(double y, double carry) {
  int lo = __double2loint(y), hi = __double2hiint(y);
  lo = __builtin_amdgcn_update_dpp(__double2loint(carry), lo, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(__double2hiint(carry), hi, 0x138, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// the same shifts with lane 0 receiving ZERO (bound_ctrl): no register has to be preloaded with the carry before every
// DPP move, which is a third of the instructions of a sweep step.  The sweeps fold the carry into lane 0's constants
// instead (the very operations a step would perform on it), see the kernels.
__device__ __forceinline__ double shr1_dz(double y) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(y), 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(y), 0x138, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float shr1_fz(float y) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float shr1_f(float y, float carry) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(carry), __float_as_int(y), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ double lane_d(double y, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), l), __builtin_amdgcn_readlane(__double2loint(y), l));
}
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ double dpp_d(double old, double y) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(y), CTRL, ROW_MASK, 0xF, BOUND);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(y), CTRL, ROW_MASK, 0xF, BOUND);
  return __hiloint2double(hi, lo);
}
// maximum over the wave of NON-NEGATIVE values (0 is the fill of the shifts): DPP moves inside the rows of 16, two row
// broadcasts across them.  (__shfl_xor is ds_bpermute: twelve dependent trips through LDS, ~2 k clocks of a lone wave per row.)
__device__ __forceinline__ double wave_max_d(double v) {
  v = fmax(v, dpp_d<0x111, 0xF, true>(0.0, v));  // row_shr:1
  v = fmax(v, dpp_d<0x112, 0xF, true>(0.0, v));  // row_shr:2
  v = fmax(v, dpp_d<0x114, 0xF, true>(0.0, v));  // row_shr:4
  v = fmax(v, dpp_d<0x118, 0xF, true>(0.0, v));  // row_shr:8   -> lane 15 of every row holds the row's maximum
  v = fmax(v, dpp_d<0x142, 0xA, false>(v, v));   // row_bcast:15 into rows 1 and 3
  v = fmax(v, dpp_d<0x143, 0xC, false>(v, v));   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the maximum
  return lane_d(v, 63);
}
// ScalarProd20 (src/hhhit-inl.h:116-122): plain left-to-right float sum of products
__device__ __forceinline__ float dot20(const float* __restrict__ q, const float* __restrict__ t) {
  float r = t[0] * q[0];
#pragma unroll
  for (int k = 1; k < 20; ++k) r = r + t[k] * q[k];
  return r;
}
__device__ __forceinline__ float fast_log2_mac(float x, const float* __restrict__ lg2, const float* __restrict__ diff) {
  if (x <= 0) return -100000;
  const uint32_t u = __float_as_uint(x);
  const int aa = (int)((u & 0x7F800000u) >> 23) - 0x7f;
  const int b = (int)((u & 0x007FE000u) >> 13);
  const int c = (int)(u & 0x00001FFFu);
  return ((float)aa + lg2[b]) + diff[b] * (float)c;
}

struct HitView {
  int Lq, Lt, pitch;
  const float* qp;
  const float* qtr;
  const float* tp;
  int tps;  // floats between consecutive template columns (20: staged profiles, 28: records of a resident template set)
  const float* ttr;
  const unsigned char* co;
  float* mat;
  unsigned char* bmm;
  double* scale;
  // secondary-structure factor fpow2(ScoreSS(q, t, i, j)) (src/hhforwardalgorithm.cpp:77,100, src/hhbackwardalgorithm.cpp:82)
  int ssm, sstw;                 // hit.ssm2 (0 = none) and the row width of its table
  const float* sstab;            // global copy of the mode's table [44*8] or [8*44]
  const unsigned char* ssq;      // [Lq+2] row index of query column i
  const unsigned char* sst;      // [Lt+2] column index of template column j; [Lt+1] = the entry the reference reads for column 1
};
__device__ __forceinline__ HitView view(const MacArgs& a, int k) {
  HitView v;
  v.Lq = a.Lq;
  v.Lt = a.Lt[k];
  v.pitch = v.Lt + 1;
  v.qp = a.q_p;
  v.qtr = a.q_tr;
  v.tps = a.t_p_stride;
  v.tp = a.t_p + (a.p_off ? a.p_off[k] : a.col_off[k]) * a.t_p_stride;
  v.ttr = a.t_tr + a.col_off[k] * 7;
  v.co = a.celloff + a.mat_off[k];
  v.mat = a.mat + a.mat_off[k];
  v.bmm = a.bmm + a.mat_off[k];
  v.scale = a.scale + (int64_t)k * (a.Lq + 2);
  v.ssm = a.ss_mode ? a.ss_mode[k] : 0;
  v.sstw = v.ssm == 1 ? 8 : 44;
  v.sstab = a.ss_tab ? a.ss_tab + (v.ssm == 2 ? 352 : 0) : nullptr;
  v.ssq = a.ss_qidx ? a.ss_qidx + (v.ssm == 2 ? (a.Lq + 2) : 0) : nullptr;
  v.sst = a.ss_tidx ? a.ss_tidx + a.ss_toff[k] : nullptr;
  return v;
}

// rows' active ranges (MacArgs::row_rng): does the strip of columns c_lo .. c_hi of a row with range r hold an active cell at all?
// The kernels that fetch mask bytes (and F_MM) from global memory strip by strip ask this first: a strip outside the range is
// masked without a trip to L2 - in a long template most strips of a row are (300 x 1800: 26 of 29).
__device__ __forceinline__ bool rng_hits(int2 r, int c_lo, int c_hi) { return c_lo <= r.y && c_hi >= r.x; }

// SPARSE rows (round 6, hits whose template has MacArgs::sparse_min_Lt columns or more): the dataflow kernels visit, in row i, only
// the strips [sa, sb] that can hold an active cell of the row OR that the next row will read (a masked cell there must read as
// zero, and the buffer holds the row before last) - a 300 x 1400 hit whose band is 120 columns wide has 3-4 such strips a row
// instead of 22.  Everything a unit reads is then either written by a visited unit of the row it reads or replaced by the zero
// it stands for (the neighbour strip's last column, see `left`); the matrix planes are cleared beforehand (mask kernel), so
// the cells nobody visits read as the reference's masked cells do.  The waves keep their DENSE unit numbers (counter value of
// unit (i, s) as if every strip were visited) and post the row's last number when they leave a row: a waiter on a unit that
// was skipped is released by the next post, and no wait changes its meaning.
struct StripSpan {
  int sa, sb;  // sa > sb: none
};
// forward geometry: strip s = columns 64 s + 1 .. 64 s + 64.  own: the row's active range; next: the next row's (empty = none)
__device__ __forceinline__ StripSpan span_fwd(bool sparse, int2 own, int2 next, int ns) {
  StripSpan r;
  r.sa = 0;
  r.sb = ns - 1;
  if (!sparse) return r;
  int lo = 0x7fffffff, hi = 0;
  if (own.x <= own.y) lo = own.x, hi = own.y;
  if (next.x <= next.y) lo = min(lo, max(1, next.x - 1)), hi = max(hi, next.y);  // row i+1 reads columns j-1 and j
  if (lo > hi) {
    r.sa = 1;
    r.sb = 0;
  } else {
    r.sa = (lo - 1) >> 6;
    r.sb = min(ns - 1, (hi - 1) >> 6);
  }
  return r;
}
// backward geometry: strip s = columns Lt - 1 - 64 s down to Lt - 64 - 64 s (column Lt is kept by the row's first step).
// next: the range of row i - 1, which reads columns j and j + 1 of this row
__device__ __forceinline__ StripSpan span_bwd(bool sparse, int2 own, int2 next, int ns, int Lt) {
  StripSpan r;
  r.sa = 0;
  r.sb = ns - 1;
  if (!sparse) return r;
  int lo = 0x7fffffff, hi = 0;
  if (own.x <= own.y) lo = own.x, hi = own.y;
  if (next.x <= next.y) lo = min(lo, next.x), hi = max(hi, next.y + 1);
  hi = min(hi, Lt - 1);
  if (lo > hi) {
    r.sa = 1;
    r.sb = 0;
  } else {
    r.sa = (Lt - 1 - hi) >> 6;
    r.sb = min(ns - 1, (Lt - 1 - lo) >> 6);
  }
  return r;
}
// RING (round 6: templates too long for the dataflow kernels' plain LDS layout, ~1450 columns): the row state lives in a ring of
// MAC_RING_STRIPS strips of 64 columns - strip s in slot s mod MAC_RING_STRIPS, a strip's columns side by side - which holds any
// row whose visited span is at most that many strips, wherever in the template it lies (sparse rows, above: a 300 x 1800 hit
// needs 3-4 strips a row).  Hits with a wider row (a large rectangle beside the alignment) stay with the single-wave kernels:
// the mask kernel leaves the widest span of the hit's rows in rng[0].x, and a workgroup of the kind that is not the hit's returns.
constexpr int MAC_RING_STRIPS = 22;                    // 22 x 64 columns x 14 doubles = 154 KB
constexpr int MAC_RING_COLS = MAC_RING_STRIPS * 64;    // slots 1 .. MAC_RING_COLS; slot MAC_RING_COLS + 1: column Lt of the backward pass
template <bool RING>
__device__ __forceinline__ int mac_col_f(int j) {  // forward geometry: strip of column j = (j - 1) >> 6
  if (!RING) return j;
  const int q = j - 1, st = q >> 6;
  return j <= 0 ? 0 : (st % MAC_RING_STRIPS) * 64 + (q & 63) + 1;
}
template <bool RING>
__device__ __forceinline__ int mac_col_b(int j, int Lt) {  // backward geometry: strip of column j <= Lt - 1 = (Lt - 1 - j) >> 6
  if (!RING) return j;
  const int q = Lt - 1 - j, st = q >> 6;
  return q < 0 ? MAC_RING_COLS + 1 : (st % MAC_RING_STRIPS) * 64 + (63 - (q & 63)) + 1;
}
// the widest row of a hit in strips (forward and backward geometry), from the rows' ranges: what decides whether the ring holds it
__device__ __forceinline__ int mac_row_span(int2 prev, int2 own, int2 next, int Lt) {
  const int nsf = (Lt + 63) >> 6, nsb = Lt >= 2 ? (Lt - 1 + 63) >> 6 : 1;
  StripSpan f, b;
  {
    int lo = 0x7fffffff, hi = 0;
    if (own.x <= own.y) lo = own.x, hi = own.y;
    if (next.x <= next.y) lo = min(lo, max(1, next.x - 1)), hi = max(hi, next.y);
    f.sa = lo > hi ? 1 : (lo - 1) >> 6;
    f.sb = lo > hi ? 0 : min(nsf - 1, (hi - 1) >> 6);
  }
  {
    int lo = 0x7fffffff, hi = 0;
    if (own.x <= own.y) lo = own.x, hi = own.y;
    if (prev.x <= prev.y) lo = min(lo, prev.x), hi = max(hi, prev.y + 1);
    hi = min(hi, Lt - 1);
    b.sa = lo > hi ? 1 : (Lt - 1 - hi) >> 6;
    b.sb = lo > hi ? 0 : min(nsb - 1, (Lt - 1 - lo) >> 6);
  }
  return max(f.sb - f.sa + 1, b.sb - b.sa + 1);
}

// the first strip >= sa that wave w of np owns (strips w, w + np, ..)
__device__ __forceinline__ int first_own(int sa, int w, int np) { return sa + (((w - sa) % np) + np) % np; }
__device__ __forceinline__ int2 rng_or_none(const MacArgs& a, int k, int i) {
  return (i >= 1 && i <= a.Lq) ? (a.row_rng + (size_t)k * (a.Lq + 2))[i] : make_int2(1, 0);
}

// Template operands of column j: from the LDS copy made at kernel start (STAGE; one hit's template is read Lq times and a
// lone wave per SIMD cannot hide a trip to L2 per strip), or from global memory when the template does not fit.
template <bool STAGE>
__device__ __forceinline__ void load_tp(const HitView& h, const float* sTp, int j, float (&t)[20]) {
  if (STAGE) {
    const float4* c = reinterpret_cast<const float4*>(sTp) + (size_t)j * 5;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const float4 v = c[q];
      t[4 * q + 0] = v.x;
      t[4 * q + 1] = v.y;
      t[4 * q + 2] = v.z;
      t[4 * q + 3] = v.w;
    }
  } else {
    const float* g = h.tp + (size_t)j * h.tps;
#pragma unroll
    for (int q = 0; q < 20; ++q) t[q] = g[q];
  }
}
template <bool STAGE>
__device__ __forceinline__ void load_tt(const HitView& h, const float* sTt, int j, float (&t)[7]) {
  if (STAGE) {
    const float4* c = reinterpret_cast<const float4*>(sTt) + (size_t)j * 2;
    const float4 a = c[0], b = c[1];
    t[0] = a.x;
    t[1] = a.y;
    t[2] = a.z;
    t[3] = a.w;
    t[4] = b.x;
    t[5] = b.y;
    t[6] = b.z;
  } else {
    const float* g = h.ttr + (size_t)j * 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) t[q] = g[q];
  }
}
// copies the template of hit h into LDS: sTp [(Lt+2)][20], sTt [(Lt+2)][8]
__device__ __forceinline__ void stage_template(const HitView& h, float* sTp, float* sTt, int lane) {
  for (int e = lane; e < (h.Lt + 1) * 20; e += 64) sTp[e] = h.tp[(size_t)(e / 20) * h.tps + (e % 20)];
  for (int e = lane; e < (h.Lt + 1) * 8; e += 64) sTt[e] = (e & 7) < 7 ? h.ttr[(size_t)(e >> 3) * 7 + (e & 7)] : 0.0f;
}
// the secondary-structure table of the hit's mode in LDS (352 floats)
__device__ __forceinline__ void stage_ss(const HitView& h, float* sSs, int lane) {
  if (h.ssm)
    for (int e = lane; e < 352; e += 64) sSs[e] = h.sstab[e];
}

// LDS row state: field f of row r at column j
#define ROW(r, f, j) rows[((r)*5 + (f)) * stride + (j)]
enum { F_MM = 0, F_GD = 1, F_IM = 2, F_DG = 3, F_MI = 4 };

}  // namespace

// ---- forward ------------------------------------------------------------------------------------------------------
// GROWS: the row state lives in global memory (a.row_scratch) instead of LDS - templates of any length.  The accesses are
// the same; rows written before a __syncthreads() are read after it, by the same (only) wavefront of the workgroup.
template <bool LOCAL, bool STAGE, bool GROWS>
__global__ void __launch_bounds__(64) hhv_mac_forward_kernel(MacArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* rows = GROWS ? a.row_scratch + (size_t)blockIdx.x * 10 * (a.lds_cols + 2) : reinterpret_cast<double*>(smem);
  const int k = a.sel[blockIdx.x], lane = threadIdx.x;
  if (GROWS && a.ring_strips > 0 && a.Lt[k] >= a.ring_min_Lt && (a.row_rng + (size_t)k * (a.Lq + 2))[0].x <= a.ring_strips) return;  // the ring kernels' hit (launch_mac_class)
  const HitView h = view(a, k);
  const int Lq = h.Lq, Lt = h.Lt, pitch = h.pitch, stride = Lt + 2;
  float* sTp = reinterpret_cast<float*>(rows + (size_t)10 * (a.lds_cols + 2));  // layout sized for the longest template
  float* sTt = sTp + (size_t)(a.lds_cols + 2) * 20;
  // STAGE: the cell-off bytes of a row are fetched from HBM while the row BEFORE it is computed (registers), parked in LDS
  // at the end of that row and read from there - a strip without active cells costs a few dozen cycles, and with the
  // fetch only one strip ahead every such strip waited a full trip to L2 / HBM for the mask byte of its successor.
  unsigned char* sCo = reinterpret_cast<unsigned char*>(sTt + (size_t)(a.lds_cols + 2) * 8);  // [2][lds_cols + 2]
  const int co_stride = a.lds_cols + 2;
  float* sSs = reinterpret_cast<float*>(sCo + (((size_t)2 * co_stride + 15) & ~(size_t)15)) + (size_t)2 * co_stride;  // [352]
  const float* sstab = STAGE ? sSs : h.sstab;
  const double Cshift = a.Cshift;
  for (int e = lane; e < 10 * stride; e += 64) rows[e] = 0.0;
  for (int e = lane; e < pitch; e += 64) h.mat[e] = 0.0f;  // row 0 of p_mm is never read
  if (STAGE) {
    stage_template(h, sTp, sTt, lane);
    stage_ss(h, sSs, lane);
    for (int j = 1 + lane; j <= Lt; j += 64) sCo[j] = h.co[(size_t)pitch + j];  // row 1
  }
  __syncthreads();
  // !STAGE: the cell-off byte of the next strip is fetched while the current one is swept (also across the row boundary).
  // Sparse rows (StripSpan; the templates the mask kernel cleared the planes of): a row visits only the strips that hold an active
  // cell of it or that the next row reads - a 300 x 1800 hit has 3-4 such strips in most rows instead of 29.  A strip left out
  // holds the row before last in the row buffer; whoever reads it there is a masked cell of a visited strip and discards it.
  const bool sparse = !STAGE && Lt >= a.sparse_min_Lt;
  const int ns = (Lt + 63) >> 6;
  int2 rr_cur = rng_or_none(a, k, 1), rr_nxt = rr_cur, rr_nn = rng_or_none(a, k, 2);  // active ranges of rows i, i + 1, i + 2 (!STAGE)
  StripSpan sp = span_fwd(sparse, rr_cur, rr_nn, ns), spn = sp;                        // strips of row i and of row i + 1
  // mask byte of this lane's column in strip ss of row ii (1 = off where the row has no active cell at all)
  auto fetch_co = [&](int ii, int ss, int2 r) -> unsigned char {
    const int jn = 1 + (ss << 6) + lane;
    return (ii <= Lq && jn <= Lt && rng_hits(r, 1 + (ss << 6), 64 + (ss << 6))) ? h.co[(size_t)ii * pitch + jn] : (unsigned char)1;
  };
  unsigned char co_next = (!STAGE && sp.sa <= sp.sb) ? fetch_co(1, sp.sa, rr_cur) : (unsigned char)1;
  int cur = 0;
  double pmin = LOCAL ? 1.0 : 0.0, scale_prod = 1.0;
  double Pf = LOCAL ? 1.0 : 0.0;
  if (lane == 0) h.scale[0] = h.scale[1] = h.scale[2] = 1.0;
  double scale_i = 1.0;  // scale[i] of the row being computed

  for (int i = 1; i <= Lq; ++i) {
    const int prv = cur ^ 1;
    if (i >= 2) {
      if (scale_prod < DBL_MIN * 100)
        scale_prod = 0.0;
      else
        scale_prod *= scale_i;
    }
    const float* qi = h.qp + (size_t)i * 20;
    const float* qt1 = h.qtr + (size_t)(i - 1) * 7;  // q.tr[i-1]
    const float* qt = h.qtr + (size_t)i * 7;         // q.tr[i]
    const double qM2M = qt1[T_M2M], qI2M = qt1[T_I2M], qD2M = qt1[T_D2M], qM2D = qt1[T_M2D], qD2D = qt1[T_D2D];
    const double qM2I = qt[T_M2I], qI2I = qt[T_I2I];
    double Pmax = 0.0, carry_mm = 0.0, carry_gd = 0.0, carry_im = 0.0;
    if (!STAGE) {
      if (i >= 2) {
        rr_cur = rr_nxt;
        rr_nxt = rr_nn;
        rr_nn = rng_or_none(a, k, i + 2);
        sp = spn;
      } else {
        rr_nxt = rr_nn;
        rr_nn = rng_or_none(a, k, 3);
      }
      spn = span_fwd(sparse, rr_nxt, rr_nn, ns);  // row i + 1
      // (a row without a strip: the mask byte of the next row's first strip is asked for here instead of in the row's last strip)
      if (sp.sa > sp.sb) co_next = (i < Lq && spn.sa <= spn.sb) ? fetch_co(i + 1, spn.sa, rr_nxt) : (unsigned char)1;
    }
    unsigned char pre_co[MAC_PRE];
    if (STAGE) {
#pragma unroll
      for (int u = 0; u < MAC_PRE; ++u) {
        const int jn = 1 + u * 64 + lane;
        pre_co[u] = (i < Lq && jn <= Lt) ? h.co[(size_t)(i + 1) * pitch + jn] : 1;
      }
    }
    const unsigned char* co_row = sCo + ((i - 1) & 1) * co_stride;
    for (int s0 = (STAGE ? 0 : sp.sa << 6); STAGE ? s0 < Lt : s0 <= (sp.sb << 6); s0 += 64) {
      const int j = 1 + s0 + lane;
      const bool valid = j <= Lt;
      const int jc = valid ? j : Lt;
      const bool off = !valid || (STAGE ? co_row[jc] != 0 : co_next != 0);
      if (!STAGE) {
        const bool last = (s0 >> 6) >= sp.sb;
        if (!last) co_next = fetch_co(i, (s0 >> 6) + 1, rr_cur);
        else co_next = (i < Lq && spn.sa <= spn.sb) ? fetch_co(i + 1, spn.sa, rr_nxt) : (unsigned char)1;
      }
      const unsigned long long on_mask = __ballot(!off);
      if (on_mask == 0) {
        // a strip without a single active cell: all five states are zero, the running sums are unchanged
        if (valid) {
          ROW(cur, F_MM, j) = 0.0;
          ROW(cur, F_GD, j) = 0.0;
          ROW(cur, F_IM, j) = 0.0;
          ROW(cur, F_DG, j) = 0.0;
          ROW(cur, F_MI, j) = 0.0;
          h.mat[(size_t)i * pitch + j] = 0.0f;
        }
        carry_mm = carry_gd = carry_im = 0.0;
        continue;
      }
      const int l0 = __builtin_ctzll(on_mask), l1 = 63 - __builtin_clzll(on_mask);
      float tpj[20], tt1[7], tt[7];
      load_tp<STAGE>(h, sTp, jc, tpj);
      load_tt<STAGE>(h, sTt, jc - 1, tt1);  // t.tr[j-1]
      load_tt<STAGE>(h, sTt, jc, tt);       // t.tr[j]
      const float pf = dot20(qi, tpj);
      // fpow2(ScoreSS(q, t, i, j)); for column 1 the reference passes (1, j) with the stale loop variable j = t.L + 1 (:77)
      float ssf = 1.0f;
      if (h.ssm && i >= 2) ssf = j == 1 ? sstab[h.ssq[1] * h.sstw + h.sst[Lt + 1]] : sstab[h.ssq[i] * h.sstw + h.sst[jc]];
      double mm, dg, mi;
      if (i == 1) {
        mm = pf * Cshift;  // :31
        dg = mi = 0.0;
      } else {
        const double pm = ROW(prv, F_MM, jc), pdg = ROW(prv, F_DG, jc), pmi = ROW(prv, F_MI, jc);
        if (j == 1) {
          mm = scale_prod * ssf * pf * Cshift;  // :71-73
        } else {
          const double m1 = ROW(prv, F_MM, jc - 1), g1 = ROW(prv, F_GD, jc - 1), i1 = ROW(prv, F_IM, jc - 1),
                       d1 = ROW(prv, F_DG, jc - 1), x1 = ROW(prv, F_MI, jc - 1);
          mm = pf * Cshift * ssf * scale_i *
               (pmin + m1 * qM2M * tt1[T_M2M] + g1 * qM2M * tt1[T_D2M] + i1 * qI2M * tt1[T_M2M] + d1 * qD2M * tt1[T_M2M] +
                x1 * qM2M * tt1[T_I2M]);  // :94-103
        }
        dg = scale_i * (pm * qM2D + pdg * qD2D);                            // :110-112 / :79-81
        mi = scale_i * (pm * qM2M * tt[T_M2I] + pmi * qM2M * tt[T_I2I]);    // :113-116 / :75-78
      }
      if (off) mm = dg = mi = 0.0;
      if (i >= 2 && j >= 2 && !off) Pmax = fmax(Pmax, mm);
      // the recurrences along the row (:104-109): gd = mm(j-1)*t[j-1][M2D] + gd(j-1)*t[j-1][D2D],
      //                                           im = mm(j-1)*q[i][M2I]*t[j-1][M2M] + im(j-1)*q[i][I2I]*t[j-1][M2M]
      const double mm_left = shr1_d(mm, carry_mm);
      const bool chain_on = !off && (i == 1 || j >= 2);  // column 1 of rows >= 2: im = gd = 0 (:74)
      const double a_gd = chain_on ? mm_left * tt1[T_M2D] : 0.0, b_gd = chain_on ? (double)tt1[T_D2D] : 0.0;
      const double c_im = chain_on ? mm_left * qM2I * tt1[T_M2M] : 0.0, b_im = chain_on ? (double)tt1[T_M2M] : 0.0;
      const double f_mm = (double)(float)mm;  // what p_mm stores (float) and Pforward sums (:168)
      // Sweep: inactive lanes hold 0 (resp. the incoming total) whatever their neighbour says, so the lanes left of the
      // first active one are final from the start and l1 - l0 + 1 steps finish everything up to the last active lane.
      // Lane 0's left neighbour is the carry of the previous strip, a constant of the sweep: its step is evaluated once,
      // with the operations of the loop body, and the loop shifts zeros into lane 0 (x + 0*b = x for the non-negative
      // finite values here).
      const bool first_lane = lane == 0;
      const double a_gd_s = first_lane ? a_gd + carry_gd * b_gd : a_gd;
      const double c_im_s = first_lane ? c_im + carry_im * qI2I * b_im : c_im;
      const double f_mm_s = first_lane ? Pf + f_mm : f_mm;
      double gd = 0.0, im = 0.0, acc = Pf;
      const int n_steps = l1 - l0 + 1;
      for (int s = 0; s < n_steps; ++s) {
        const double gl = shr1_dz(gd), il = shr1_dz(im);
        gd = a_gd_s + gl * b_gd;
        im = c_im_s + il * qI2I * b_im;
        if (LOCAL) acc = shr1_dz(acc) + f_mm_s;
      }
      if (valid) {
        ROW(cur, F_MM, j) = mm;
        ROW(cur, F_GD, j) = gd;
        ROW(cur, F_IM, j) = im;
        ROW(cur, F_DG, j) = dg;
        ROW(cur, F_MI, j) = mi;
        h.mat[(size_t)i * pitch + j] = (float)mm;
      }
      carry_mm = lane_d(mm, 63);
      carry_gd = lane_d(gd, 63);   // lane 63 is either inactive (0) or l1 (final)
      carry_im = lane_d(im, 63);
      if (LOCAL) Pf = lane_d(acc, l1);  // lanes right of l1 would only add zeros
    }
    if (lane == 0) h.mat[(size_t)i * pitch] = 0.0f;
    if (STAGE) {
      unsigned char* co_nextrow = sCo + (i & 1) * co_stride;
#pragma unroll
      for (int u = 0; u < MAC_PRE; ++u) {
        const int jn = 1 + u * 64 + lane;
        if (jn <= Lt) co_nextrow[jn] = pre_co[u];
      }
    }
    double scale_next = 1.0;
    if (i >= 2) {
      Pmax = wave_max_d(Pmax);
      pmin *= scale_i;
      if (pmin < DBL_MIN * 100) pmin = 0.0;
      scale_next = 1.0 / (Pmax + 1.0);  // :155
      if (lane == 0) h.scale[i + 1] = scale_next;
    }
    __syncthreads();
    // total forward probability (:162-182)
    if (LOCAL) {
      Pf *= scale_next;
    } else if (i < Lq) {
      // (a sparse row's last strip need not be the template's: F_MM of column Lt is then 0, the buffer holds an older row)
      const double fL = (STAGE || (sp.sa <= sp.sb && sp.sb == ns - 1)) ? ROW(cur, F_MM, Lt) : 0.0;
      Pf = (Pf + (float)fL * scale_next);
    }
    scale_i = scale_next;
    cur = prv;
  }
  if (!LOCAL) {
    // + sum_j F(Lq, j) in column order, then * scale[Lq+1]
    const int last = cur ^ 1;
    for (int s0 = 0; s0 < Lt; s0 += 64) {
      const int j = 1 + s0 + lane;
      const bool seen = STAGE || ((s0 >> 6) >= sp.sa && (s0 >> 6) <= sp.sb);  // (strips row Lq did not visit hold an older row)
      const double f_mm = (j <= Lt && seen) ? (double)(float)ROW(last, F_MM, j) : 0.0;
      double acc = Pf;
      for (int s = 0; s < 64; ++s) acc = shr1_d(acc, Pf) + f_mm;
      Pf = lane_d(acc, 63);
    }
    Pf *= scale_i;
  }
  if (lane == 0) a.Pforward[k] = Pf;
}

// ---- forward list (-o_matrices) ---------------------------------------------------------------------------------------
// src/hhforwardalgorithm.cpp:184-219, run between forward and backward (backward turns F_MM into the posterior in place):
// ffprob = F_MM(i, j) / Pforward * scale_rate(i) in double; where the reference pushes an entry (ffprob > 1e-4) the dense plane
// a.fwd_list takes (float)ffprob, elsewhere 0.  One workgroup per hit; every thread carries the row constants itself.
__global__ void __launch_bounds__(256) hhv_mac_fwdlist_kernel(MacArgs a) {
  const int k = a.sel[blockIdx.x];
  const HitView h = view(a, k);
  const int Lq = h.Lq, Lt = h.Lt, pitch = h.pitch;
  float* out = a.fwd_list + a.mat_off[k];
  const double Pf = a.Pforward[k];
  double scale_prod = 1.0;  // as the forward loop leaves it (:17, :66-69)
  for (int i = 2; i <= Lq; ++i) {
    if (scale_prod < DBL_MIN * 100)
      scale_prod = 0.0;
    else
      scale_prod *= h.scale[i];
  }
  const double top = scale_prod * h.scale[Lq + 1];
  double scale_prod_curr = 1.0;
  for (int i = 1; i <= Lq; ++i) {
    if (scale_prod_curr < DBL_MIN * 100)
      scale_prod_curr = 0.0;
    else
      scale_prod_curr *= h.scale[i];
    const double scale_rate = scale_prod_curr == 0.0 ? 0.0 : top / scale_prod_curr;
    for (int j = 1 + (int)threadIdx.x; j <= Lt; j += 256) {
      const double ff = ((double)h.mat[(size_t)i * pitch + j] / Pf) * scale_rate;
      out[(size_t)i * pitch + j] = ff > (double)MAC_LIST_THRESHOLD ? (float)ff : 0.0f;
    }
  }
}

// ---- backward + posterior ---------------------------------------------------------------------------------------------
// LISTS: the values of the reference's sparse backward list (-o_matrices; src/hhbackwardalgorithm.cpp:112-122) go into the
// dense plane a.bwd_list: the value where the reference would push an entry, 0 elsewhere (an entry's value is > 1e-4).
template <bool LOCAL, bool STAGE, bool GROWS, bool LISTS>
__global__ void __launch_bounds__(64) hhv_mac_backward_kernel(MacArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* rows = GROWS ? a.row_scratch + (size_t)blockIdx.x * 10 * (a.lds_cols + 2) : reinterpret_cast<double*>(smem);
  const int k = a.sel[blockIdx.x], lane = threadIdx.x;
  if (GROWS && a.ring_strips > 0 && a.Lt[k] >= a.ring_min_Lt && (a.row_rng + (size_t)k * (a.Lq + 2))[0].x <= a.ring_strips) return;  // the ring kernels' hit (launch_mac_class)
  const HitView h = view(a, k);
  const int Lq = h.Lq, Lt = h.Lt, pitch = h.pitch, stride = Lt + 2;
  float* sTp = reinterpret_cast<float*>(rows + (size_t)10 * (a.lds_cols + 2));
  float* sTt = sTp + (size_t)(a.lds_cols + 2) * 20;
  // STAGE: mask bytes and F_MM of a row are fetched while the row processed before it is computed (see the forward kernel)
  unsigned char* sCo = reinterpret_cast<unsigned char*>(sTt + (size_t)(a.lds_cols + 2) * 8);  // [2][lds_cols + 2]
  const int co_stride = a.lds_cols + 2;
  float* sF = reinterpret_cast<float*>(sCo + (((size_t)2 * co_stride + 15) & ~(size_t)15));  // [2][lds_cols + 2]
  float* sSs = sF + (size_t)2 * co_stride;                                                    // [352]
  const float* sstab = STAGE ? sSs : h.sstab;
  const double Cshift = a.Cshift, Pf = a.Pforward[k];
  for (int e = lane; e < 10 * stride; e += 64) rows[e] = 0.0;
  if (STAGE) {
    stage_template(h, sTp, sTt, lane);
    stage_ss(h, sSs, lane);
    if (Lq >= 2)
      for (int j = 1 + lane; j <= Lt; j += 64) {  // row Lq - 1, the first one of the loop below
        sCo[((Lq - 1) & 1) * co_stride + j] = h.co[(size_t)(Lq - 1) * pitch + j];
        sF[((Lq - 1) & 1) * co_stride + j] = h.mat[(size_t)(Lq - 1) * pitch + j];
      }
  }
  __syncthreads();
  const double sL = h.scale[Lq + 1];
  int cur = 0;  // row being computed; `prv` holds row i+1
  // row Lq (:19-29)
  for (int j = 1 + lane; j <= Lt; j += 64) {
    float* pv = h.mat + (size_t)Lq * pitch + j;
    if (h.co[(size_t)Lq * pitch + j]) {
      *pv = 0.0f;
      ROW(1, F_MM, j) = 0.0;
    } else {
      ROW(1, F_MM, j) = sL;
      *pv = (float)(*pv * sL / Pf);
    }
  }
  __syncthreads();
  double scale_prod = sL, pmin = LOCAL ? sL : 0.0;
  double final_scale_prod = sL;  // :31-36
  if (LISTS) {
    for (int i = Lq - 1; i >= 1; --i) {
      final_scale_prod *= h.scale[i + 1];
      if (final_scale_prod < DBL_MIN * 100) final_scale_prod = 0.0;
    }
  }
  float* blist = LISTS ? a.bwd_list + a.mat_off[k] : nullptr;
  unsigned char co_nx = 1;  // mask byte and F_MM of the next strip, fetched while the current one is swept
  float f_nx = 0.0f;
  // sparse rows as in the forward kernel (StripSpan, backward geometry); a hit whose Pforward is not a positive number keeps every
  // strip: the reference's posterior is then NaN in EVERY cell, which only a visit writes
  const bool sparse = !STAGE && Lt >= a.sparse_min_Lt && Pf > 0.0 && Pf < 1.0e300;
  const int nsb = Lt >= 2 ? (Lt - 1 + 63) >> 6 : 0;
  int2 rr_row = make_int2(1, 0), rr_nxt = rng_or_none(a, k, Lq - 1);
  for (int i = Lq - 1; i >= 1; --i) {
    const int prv = cur ^ 1;
    const double sc = h.scale[i + 1];
    scale_prod *= sc;
    if (scale_prod < DBL_MIN * 100) scale_prod = 0.0;
    pmin *= sc;
    if (pmin < DBL_MIN * 100) pmin = 0.0;
    float* row = h.mat + (size_t)i * pitch;
    const unsigned char* corow = h.co + (size_t)i * pitch;
    unsigned char pre_co[MAC_PRE];
    float pre_f[MAC_PRE];
    StripSpan sp;
    sp.sa = 0;
    sp.sb = nsb - 1;
    const unsigned char* co_l = sCo + (i & 1) * co_stride;
    const float* f_l = sF + (i & 1) * co_stride;
    if (STAGE) {
#pragma unroll
      for (int u = 0; u < MAC_PRE; ++u) {
        const int jn = 1 + u * 64 + lane;
        const bool in = i >= 2 && jn <= Lt;
        pre_co[u] = in ? h.co[(size_t)(i - 1) * pitch + jn] : 1;
        pre_f[u] = in ? h.mat[(size_t)(i - 1) * pitch + jn] : 0.0f;
      }
    } else {
      // what the first strip of this row reads from HBM (mask byte, F_MM) and, lane 0, column Lt - issued together
      // (a strip outside the row's active range: masked, and F_MM is the 0 the forward pass left there)
      rr_row = rr_nxt;
      rr_nxt = rng_or_none(a, k, i - 1);  // (a row ahead: a scalar load at the start of every row would be waited for)
      sp = span_bwd(sparse, rr_row, rr_nxt, nsb, Lt);
      const int j0 = Lt - 1 - (sp.sa << 6) - lane;
      const bool live = sp.sa <= sp.sb && rng_hits(rr_row, j0 + lane - 63, j0 + lane);
      co_nx = (j0 >= 1 && live) ? corow[j0] : 1;
      f_nx = (j0 >= 1 && live) ? row[j0] : 0.0f;
    }
    // column Lt (:58-71)
    if (lane == 0) {
      const unsigned char coL = STAGE ? co_l[Lt] : corow[Lt];
      const float fL = STAGE ? f_l[Lt] : row[Lt];
      if (coL) {
        row[Lt] = 0.0f;
        ROW(cur, F_MM, Lt) = 0.0;
      } else {
        ROW(cur, F_MM, Lt) = scale_prod;
        row[Lt] = (float)(fL * scale_prod / Pf);
      }
      ROW(cur, F_GD, Lt) = ROW(cur, F_IM, Lt) = ROW(cur, F_DG, Lt) = ROW(cur, F_MI, Lt) = 0.0;
    }
    const float* qn = h.qp + (size_t)(i + 1) * 20;
    const float* qt = h.qtr + (size_t)i * 7;
    const double qM2M = qt[T_M2M], qM2I = qt[T_M2I], qM2D = qt[T_M2D], qI2M = qt[T_I2M], qI2I = qt[T_I2I], qD2M = qt[T_D2M],
                 qD2D = qt[T_D2D];
    double carry_gd = 0.0, carry_im = 0.0;  // curr[Lt].gd = curr[Lt].im = 0
    for (int s0 = sp.sa << 6; s0 <= (sp.sb << 6); s0 += 64) {
      const int j = Lt - 1 - s0 - lane;  // descending: lane 0 is the rightmost column of the strip
      const bool valid = j >= 1;
      const int jc = valid ? j : 1;
      const bool off = !valid || (STAGE ? co_l[jc] != 0 : co_nx != 0);
      const float f_cur = STAGE ? (valid ? f_l[jc] : 0.0f) : f_nx;
      if (!STAGE && (s0 >> 6) < sp.sb) {
        const int jn = j - 64;
        const bool live = rng_hits(rr_row, jn + lane - 63, jn + lane);
        co_nx = (jn >= 1 && live) ? corow[jn] : 1;
        f_nx = (jn >= 1 && live) ? row[jn] : 0.0f;
      }
      const unsigned long long on_mask = __ballot(!off);
      if (on_mask == 0) {
        if (valid) {
          ROW(cur, F_MM, j) = 0.0;
          ROW(cur, F_GD, j) = 0.0;
          ROW(cur, F_IM, j) = 0.0;
          ROW(cur, F_DG, j) = 0.0;
          ROW(cur, F_MI, j) = 0.0;
          row[j] = f_cur * (float)(0.0 / Pf);  // F * (float)(B / Pforward) with B = 0 (NaN if Pforward is 0, as in the reference)
        }
        carry_gd = carry_im = 0.0;
        continue;
      }
      const int l0 = __builtin_ctzll(on_mask), l1 = 63 - __builtin_clzll(on_mask);
      float tpn[20], tt[7];
      load_tp<STAGE>(h, sTp, jc + 1, tpn);
      load_tt<STAGE>(h, sTt, jc, tt);
      const float pf = dot20(qn, tpn);
      const float ssf = h.ssm ? sstab[h.ssq[i + 1] * h.sstw + h.sst[jc + 1]] : 1.0f;  // fpow2(ScoreSS(q, t, i+1, j+1))
      const double pmatch = ROW(prv, F_MM, jc + 1) * pf * ssf * Cshift * sc;  // :80-83
      const double pdg = ROW(prv, F_DG, jc), pmi = ROW(prv, F_MI, jc);
      const double tM2M = tt[T_M2M];
      double dg = (+pmatch * qD2M * tM2M + pdg * qD2D * sc);                       // :103-106
      double mi = (+pmatch * qM2M * tt[T_I2M] + pmi * qM2M * tt[T_I2I] * sc);     // :108-111
      const double a_gd = off ? 0.0 : pmatch * qM2M * tt[T_D2M], b_gd = off ? 0.0 : (double)tt[T_D2D];  // :95-97
      const double c_im = off ? 0.0 : pmatch * qI2M * tM2M, b_im = off ? 0.0 : tM2M;                    // :99-101
      const bool first_lane = lane == 0;  // its neighbour is the carry of the previous strip: folded, see the forward kernel
      const double a_gd_s = first_lane ? a_gd + carry_gd * b_gd : a_gd;
      const double c_im_s = first_lane ? c_im + carry_im * qI2I * b_im : c_im;
      double gd = 0.0, im = 0.0;
      const int n_steps = l1 - l0 + 1;
      for (int s = 0; s < n_steps; ++s) {
        const double gr = shr1_dz(gd), ir = shr1_dz(im);
        gd = a_gd_s + gr * b_gd;
        im = c_im_s + ir * qI2I * b_im;
      }
      const double gr = shr1_d(gd, carry_gd), ir = shr1_d(im, carry_im);  // curr[j+1].gd / .im
      double mm = (+pmin + pmatch * qM2M * tM2M + gr * tt[T_M2D] + ir * qM2I * tM2M + pdg * qM2D * sc +
                   pmi * qM2M * tt[T_M2I] * sc);  // :86-93
      if (off) mm = dg = mi = 0.0;
      if (valid) {
        ROW(cur, F_MM, j) = mm;
        ROW(cur, F_GD, j) = gd;
        ROW(cur, F_IM, j) = im;
        ROW(cur, F_DG, j) = dg;
        ROW(cur, F_MI, j) = mi;
        row[j] = f_cur * (float)(mm / Pf);  // multiplyPosteriorValue(i, jj, float) (:122-124)
      }
      if (LISTS && valid && !off) {
        // :112-122: float = ProbFwd(q.p[i], t.p[j]) * Cshift * B_MM / Pforward * final_scale_prod / scale_prod, left to right
        float tpj[20];
        load_tp<STAGE>(h, sTp, jc, tpj);
        const float sub = dot20(h.qp + (size_t)i * 20, tpj);
        const float v = (float)((double)sub * Cshift * mm / Pf * final_scale_prod / scale_prod);
        if (v > MAC_LIST_THRESHOLD) blist[(size_t)i * pitch + j] = v;
      }
      carry_gd = lane_d(gd, 63);
      carry_im = lane_d(im, 63);
    }
    if (STAGE) {
      unsigned char* co_n = sCo + ((i - 1) & 1) * co_stride;
      float* f_n = sF + ((i - 1) & 1) * co_stride;
#pragma unroll
      for (int u = 0; u < MAC_PRE; ++u) {
        const int jn = 1 + u * 64 + lane;
        if (jn <= Lt) {
          co_n[jn] = pre_co[u];
          f_n[jn] = pre_f[u];
        }
      }
    }
    __syncthreads();
    cur = prv;
  }
}

// ---- forward / backward as a dataflow of the wavefronts of one workgroup (round 5) ---------------------------------------
// A lone wave is latency-bound in BOTH halves of a unit (row, strip of 64 columns): the parallel part - operand loads, a 20-term
// dot product, some thirty fp64 operations per cell, every step waiting for the one before - takes ~2 k clocks, and a sweep step of
// the IM chain is shift -> multiply -> multiply -> add, four dependent instructions (60 clocks a step, tools/mac_chain_ubench.hip).
// The dependencies leave room: the parallel part of (row i, strip s) needs the chains of (row i-1, strips s and s-1) and - forward
// only - the rescaling factor of row i-1, i.e. max_j F_MM over ALL strips of row i-1, which comes out of the parallel part alone;
// the units of one row are independent of each other but for one value (forward: F_MM of the column left of the strip).  So a hit
// gets a workgroup of seven (backward: eight) wavefronts, each running its own loop over its units and waiting on progress
// counters in LDS (monotone, one per wave; LDS executes a workgroup's operations in order, so a counter written after the data is
// seen after it; the counters are read and written with explicit ds instructions, see df_wait):
//   P   (waves 0 .. MAC_NP-1)  the parallel part of the strips s = w, w + MAC_NP, ..: MM, DG, MI; the additive terms of the chains
//                    into the slots the results will take (ROW(cur, F_GD / F_IM, j)), their factors into XB(0 / 1, j), the
//                    Pforward summand into XB(2, j), the strip's mask of active lanes.  Waits for the chains (and the total) of the
//                    unit above; backward it also stages the next row's mask bytes and F_MM values for everybody.
//   chains (4 waves) the GD / IM recurrences of the rows of one parity each, WALKED by one lane (mac_walk: 14 / 21 clocks a column
//                    against 60 a sweep step); wait for P of their unit.  Two rows' chains are in flight.
//   ST  (forward)    the running total of Pforward over all units in order (its own chain through the whole matrix), and the
//                    global stores of F_MM: a wave that stores never waits for a load of its own
//   P2  (backward, 2 waves: even / odd strips)  B_MM from the finished chains (curr[j+1].gd / .im), posted to the row below at
//                    once, then the posterior F*B/Pforward and the -omat list entry; backward P waits for P2 of the unit above.
// Every value is computed by the operations of the single-wave kernels in their order, so the results are the same bits (the GPU
// suite and the soak compare them with the reference).  Two row buffers are enough: a buffer is rewritten by P two rows later, and
// P has by then waited for every reader of the old row (argued at each wait below).
// Measured (500 hits 300 x 300, one session): forward 1.84 -> 1.32 ms, backward 1.55 -> 1.37 ms; a launch with at most one hit per
// CU: 1.14 ms.  What was tried on the way and did not pay: a lockstep pipeline (barriers per phase: 1.72 ms), sweeps instead of
// walks in the chain waves, four P waves (no faster alone, and two such workgroups on a CU ran at half speed): NOTES_r5.md.
// -DHHV_MAC_TIMING (make lib_variant NAME=mt FLAGS=-DHHV_MAC_TIMING) records when the waves of workgroup 0 pass their waits and
// post their units for two rows and prints the table at the end of the kernel.
// Row state in global memory (GROWS) stays with the single-wave kernels.
constexpr int MAC_ROW_FIELDS = 14;    // two rows of five states + XB(0..3)
constexpr int MAC_DF_STRIPS = 24;     // strips per row the mask table holds (LDS limits the templates of these kernels to ~1420 columns)
constexpr int MAC_CTL_DOUBLES = 64;   // behind the rows: masks [2][24] (48), per-row rings (forward 2 + 2 x MAC_NP, backward 8), 12 counters (6)
static_assert(2 * MAC_DF_STRIPS + 2 + 2 * 4 + 6 <= MAC_CTL_DOUBLES, "control block");
// (the dataflow kernels address columns through DFC: the plain layout or the ring, mac_col_f / mac_col_b)
#undef ROW
#define ROW(r, f, j) rows[((r)*5 + (f)) * stride + DFC(j)]
#define XB(k, j) rows[(10 + (k)) * stride + DFC(j)]
#define MSK(s) (RING ? (s) % MAC_RING_STRIPS : (s))
// Wavefronts of the parallel part (wave w works on strips w, w + MAC_NP, .. of every row): as many as keep the workgroup at EIGHT
// wavefronts.  The kernels need ~100 VGPRs, i.e. four waves per SIMD, sixteen per CU: two workgroups of eight share a CU (500 hits on
// 256 CUs), two of nine or ten do not - measured: four P waves forward 1.27 -> 1.8 ms, three backward 1.26 -> 1.63 ms.
#define MAC_NP_FWD 3  // + four chain waves + the total
#define MAC_NP_BWD 2  // + four chain waves + two posterior waves
#define MAC_OWN ((13 + MAC_NP - 1) / MAC_NP)  // strips of a row one P wave owns at most in the staged classes (MAC_PRE / MAC_NP, rounded up)
enum { DF_P = 0 /* [MAC_NP] */, DF_S = 4 /* [parity][chain] */, DF_T = 8 /* [2] */, DF_R = 10, DF_DEAD = 11, DF_N = 12 };
constexpr int MAC_DF_THREADS = (MAC_NP_FWD + 5) * 64;
constexpr int MAC_DFB_THREADS = (MAC_NP_BWD + 6) * 64;
#define MAC_NP MAC_NP_FWD

// progress counters: wave uniform, written by one lane
#if defined(HHV_MAC_TIMING)  // measurement build (make lib_variant NAME=mt FLAGS=-DHHV_MAC_TIMING): where the waves of workgroup 0 wait
#define DF_TIMING_DECL unsigned long long t_wait = 0, t_begin = __builtin_readcyclecounter(), ev_t[32]; int n_wait = 0, ev_n = 0, ev_c[32];
constexpr int DF_EV_ROW = 150;
#define DF_EVENT(code, i, s) if (blockIdx.x == 0 && (i) >= DF_EV_ROW && (i) <= DF_EV_ROW + 1 && ev_n < 32) { ev_t[ev_n] = __builtin_readcyclecounter(); ev_c[ev_n++] = (code) * 10000 + (i) * 10 + (s); }
#define DF_TIMING_REPORT(name)                                                                                              \
  if (blockIdx.x == 0 && lane == 0)                                                                                         \
    printf("%s wave %d: total %llu clk, waiting %llu clk in %d waits that did not pass at once\n", name, wv,                 \
           (unsigned long long)(__builtin_readcyclecounter() - t_begin), t_wait, n_wait);                                   \
  if (blockIdx.x == 0 && lane == 0)                                                                                         \
    for (int e_ = 0; e_ < ev_n; ++e_) printf("EV %s w%d code %d t %llu\n", name, wv, ev_c[e_], ev_t[e_]);
#define DF_WAIT(...) { const unsigned long long t0_ = __builtin_readcyclecounter(); if (df_wait(__VA_ARGS__)) { ++n_wait; t_wait += __builtin_readcyclecounter() - t0_; } }
#else
#define DF_TIMING_DECL
#define DF_EVENT(code, i, s)
#define DF_TIMING_REPORT(name)
#define DF_WAIT(...) df_wait(__VA_ARGS__)
#endif
// The counters are addressed as LDS explicitly (ds_read / ds_write through inline assembly): through a volatile generic pointer
// the compiler emits FLAT accesses, whose completion is counted with the global-memory counter as well - every poll then waited
// for the wave's outstanding global loads and stores (s_waitcnt vmcnt(0) in the wait loop, ~3-5 k clocks at every row boundary).
struct LdsCnt {
  uint32_t a;  // LDS byte address of a 32-bit counter (0 = none)
  __device__ __forceinline__ LdsCnt operator+(int k) const { return LdsCnt{a + 4u * (uint32_t)k}; }
};
__device__ __forceinline__ LdsCnt lds_cnt(const void* p) { return LdsCnt{(uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p}; }
#if defined(HHV_EXP_MAC_TIMEOUT)  // TEST build (make lib_pto, tests/test_gpu_errors.py): wave 0 never posts, the others give up at once
constexpr int DF_WAIT_SPINS = 1 << 6;
#else
constexpr int DF_WAIT_SPINS = 1 << 18;  // x (a poll + s_sleep 1): tens of milliseconds
#endif
// up to three counters at once: their reads go out together - one trip to LDS instead of three.
// Bounded: a wave that never sees its counter gives up (the results are then wrong and the parity tests say so) instead of
// hanging the device; after the first time-out nobody waits any more.  Returns whether it had to wait.
__device__ __forceinline__ bool df_wait(LdsCnt c, int need, LdsCnt dead, LdsCnt c2 = LdsCnt{0}, int need2 = 0, LdsCnt c3 = LdsCnt{0},
                                        int need3 = 0) {
  const uint32_t a2 = c2.a ? c2.a : c.a, a3 = c3.a ? c3.a : c.a;
  if (!c2.a) need2 = need;
  if (!c3.a) need3 = need;
  bool waited = false;
  for (int g = 0; g < DF_WAIT_SPINS; ++g) {
    // (one lane polls: with all 64 reading, the polls of a dozen waiting waves took half of the CU's LDS bandwidth from the
    // waves that work - two hits on a CU ran as slowly as one after the other)
    int ok = 0;
    if ((threadIdx.x & 63) == 0) {
      int v1, v2, v3, d;
      asm volatile(
          "ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
          : "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(d)
          : "v"(c.a), "v"(a2), "v"(a3), "v"(dead.a)
          : "memory");
      ok = ((v1 >= need && v2 >= need2 && v3 >= need3) || d != 0) ? 1 : 0;
    }
    asm volatile("" ::: "memory");
    if (__builtin_amdgcn_readlane(ok, 0)) break;
    if (g == DF_WAIT_SPINS - 1) asm volatile("ds_write_b32 %0, %1" ::"v"(dead.a), "v"(1) : "memory");
    waited = true;
    __builtin_amdgcn_s_sleep(1);
  }
  // (everything the waves hand each other lives in LDS, which executes the operations of a workgroup in order: the compiler
  // barrier of the asm statements is all that is needed - no fence, which would wait for this wave's global stores too)
  return waited;
}
// a wave that gave up: the results of the hit are wrong - say so in the context's error word (every call that waits for the
// stream then answers HHV_E_DEVICE), once per wave at the end of the kernel
__device__ __forceinline__ void df_report(LdsCnt dead, uint32_t* err, int lane) {
  int d;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(d) : "v"(dead.a) : "memory");
  if (d != 0 && lane == 0 && err) __hip_atomic_fetch_or(err, DEV_ERR_MAC_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void df_post(LdsCnt c, int v, int lane) {
  if (lane == 0)
    asm volatile("ds_write_b32 %0, %1" ::"v"(c.a), "v"(v) : "memory");
  else
    asm volatile("" ::: "memory");
}

// A chain WALKED by one lane instead of swept by 64 (tools/mac_chain_ubench.hip, one wave per SIMD: a sweep step costs 60 clocks for
// the IM chain, 63 for GD, 45 for the total - the DPP move behind an fp64 result is slow - while the bare dependent fp64 operations
// of a column take 21, 14 and 7).  The lane reads the operands the parallel part left in LDS four columns ahead of their use:
//   KIND 0  y = c[e] + y * b[e]          (GD)       the result replaces c[e]
//   KIND 1  y = c[e] + (y * q) * b[e]    (IM)       the result replaces c[e]
//   KIND 2  y = y + c[e]                 (the running total of Pforward)
// over n columns e = 0 .. n-1 at element stride DIR (backward walks towards smaller columns), the reference's operations in the
// reference's order (src/hhforwardalgorithm.cpp:104-109,168, src/hhbackwardalgorithm.cpp:95-101).
template <int DIR, int KIND>
__device__ __forceinline__ double mac_walk(double* __restrict__ pc, const double* __restrict__ pb, int n, double y, double q) {
  // blocks of four columns in three register sets: a block's operands are requested EIGHT columns (two blocks) before their use
  // (one block ahead - ~90 clocks of fp64 chain - is less than a trip to LDS under load).  Measured against one block ahead:
  // 256 hits 3.29 -> 3.26 ms, 500 hits unchanged - the walk's 48-60 clocks a column in place (21 in isolation) are not the operand
  // latency alone (NOTES_r5 section 6)
  double cA[4], bA[4], cB[4], bB[4], cC[4], bC[4];
#define MAC_WALK_LOAD(X, E)                       \
  _Pragma("unroll") for (int u = 0; u < 4; ++u) { \
    c##X[u] = pc[((E) + u) * DIR];                \
    if (KIND != 2) b##X[u] = pb[((E) + u) * DIR]; \
  }
#define MAC_WALK_STEP(C, B, E)                       \
  {                                                  \
    if (KIND == 0) {                                 \
      y = (C) + y * (B);                             \
    } else if (KIND == 1) {                          \
      double t = y * q;                              \
      t = t * (B);                                   \
      y = (C) + t;                                   \
    } else {                                         \
      y = y + (C);                                   \
    }                                                \
    if (KIND != 2) pc[(E) * DIR] = y;                \
  }
#define MAC_WALK_BLOCK(X, E) _Pragma("unroll") for (int u = 0; u < 4; ++u) MAC_WALK_STEP(c##X[u], b##X[u], (E) + u)
  int e = 0;
  if (n >= 4) { MAC_WALK_LOAD(A, 0) }
  if (n >= 8) { MAC_WALK_LOAD(B, 4) }
  // at the top: A holds block e (if it is a whole block), B block e + 4
  while (e + 12 <= n) {
    MAC_WALK_LOAD(C, e + 8)
    MAC_WALK_BLOCK(A, e)
    if (e + 16 <= n) { MAC_WALK_LOAD(A, e + 12) }
    MAC_WALK_BLOCK(B, e + 4)
    if (e + 20 <= n) { MAC_WALK_LOAD(B, e + 16) }
    MAC_WALK_BLOCK(C, e + 8)
    e += 12;
  }
  if (e + 4 <= n) {
    MAC_WALK_BLOCK(A, e)
    e += 4;
    if (e + 4 <= n) {
      MAC_WALK_BLOCK(B, e)
      e += 4;
    }
  }
  // the last one to three columns: their operands in one go
  const int r = n - e;
  if (r > 0) {
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (u < r) {
        cC[u] = pc[(e + u) * DIR];
        if (KIND != 2) bC[u] = pb[(e + u) * DIR];
      }
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (u < r) MAC_WALK_STEP(cC[u], bC[u], e + u)
  }
#undef MAC_WALK_BLOCK
#undef MAC_WALK_LOAD
#undef MAC_WALK_STEP
  return y;
}

// The query row's constants of a P wave, fetched a ROW AHEAD: lane l < 20 holds q.p[row][l], lanes 20.. the transitions the
// row needs; at the start of the row the values move to scalars with readlane (a scalar load at the start of every row cost the
// wave a trip to L2 with nothing else to do).
__device__ __forceinline__ float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ void stage_template_wg(const HitView& h, float* sTp, float* sTt, int tid, int nt) {
  for (int e = tid; e < (h.Lt + 1) * 20; e += nt) sTp[e] = h.tp[(size_t)(e / 20) * h.tps + (e % 20)];
  for (int e = tid; e < (h.Lt + 1) * 8; e += nt) sTt[e] = (e & 7) < 7 ? h.ttr[(size_t)(e >> 3) * 7 + (e & 7)] : 0.0f;
}

#define DFC(j) mac_col_f<RING>(j)
template <bool LOCAL, bool STAGE, bool RING = false>
__global__ void __launch_bounds__(MAC_DF_THREADS) hhv_mac_forward_df_kernel(MacArgs a) {
  static_assert(!(RING && STAGE), "the ring is for templates that do not fit LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* rows = reinterpret_cast<double*>(smem);
  constexpr int NT = MAC_DF_THREADS;
  const int k = a.sel[blockIdx.x], tid = threadIdx.x, lane = tid & 63;
  if (RING && (a.Lt[k] < a.ring_min_Lt || (a.row_rng + (size_t)k * (a.Lq + 2))[0].x > MAC_RING_STRIPS)) return;  // a short template in this class for capacity, or a row wider than the ring: the single-wave kernels
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  DF_TIMING_DECL
  const HitView h = view(a, k);
  const int Lq = h.Lq, Lt = h.Lt, pitch = h.pitch, stride = RING ? MAC_RING_COLS + 2 : Lt + 2;
  const size_t cols = (size_t)a.lds_cols + 2;  // layout sized for the longest template of the launch
  double* ctl = rows + MAC_ROW_FIELDS * cols;
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(ctl);  // [2][MAC_DF_STRIPS] active lanes of (row & 1, strip)
  double* rring = ctl + 2 * MAC_DF_STRIPS;                                 // [2] scale[i+1] of row i
  double* pmaxring = rring + 2;                                            // [2][MAC_NP] max_j F_MM of row i over the strips of a P wave
  int* cnt_mem = reinterpret_cast<int*>(pmaxring + 2 * MAC_NP);
  const LdsCnt cnt = lds_cnt(cnt_mem);
  float* sTp = reinterpret_cast<float*>(ctl + MAC_CTL_DOUBLES);
  float* sTt = sTp + cols * 20;
  unsigned char* sCo = reinterpret_cast<unsigned char*>(sTt + cols * 8);  // [2][cols]: the mask bytes of a row, fetched a row ahead (P only)
  const int co_stride = (int)cols;
  float* sSs = reinterpret_cast<float*>(sCo + (((size_t)2 * co_stride + 15) & ~(size_t)15)) + (size_t)2 * co_stride;  // [352]
  const float* sstab = STAGE ? sSs : h.sstab;
  const double Cshift = a.Cshift;
  for (int e = tid; e < 10 * stride; e += NT) rows[e] = 0.0;
  for (int e = tid; e < pitch; e += NT) h.mat[e] = 0.0f;  // row 0 of p_mm is never read
  if (STAGE) {
    stage_template_wg(h, sTp, sTt, tid, NT);
    if (h.ssm)
      for (int e = tid; e < 352; e += NT) sSs[e] = h.sstab[e];
    for (int j = 1 + tid; j <= Lt; j += NT) sCo[co_stride + j] = h.co[(size_t)pitch + j];  // row 1 (buffer row & 1)
  }
  if (tid < DF_N) cnt_mem[tid] = 0;
  if (tid < 2 * MAC_NP) pmaxring[tid] = 0.0;
  if (tid == 0) h.scale[0] = h.scale[1] = h.scale[2] = 1.0;
  __syncthreads();
  const int ns = (Lt + 63) >> 6;
  const bool sparse = RING || Lt >= a.sparse_min_Lt;  // (see StripSpan)
  // strips of a row that P wave w works on
#define N_OF(w) (ns > (w) ? (ns - (w) + MAC_NP - 1) / MAC_NP : 0)
  const LdsCnt dead = cnt + DF_DEAD;
  // unit (i, s) of the parallel part is done when its wave's counter has reached ...
  // (a row takes N_OF + 1 numbers of a P wave's counter: its units, then "row finished" - the row's maximum is published between the
  // last unit and that number, and the number of the last unit must not release those who wait for the maximum)
#define P_DONE(i, s) cnt + DF_P + ((s) % MAC_NP), ((i)-1) * (N_OF((s) % MAC_NP) + 1) + (s) / MAC_NP + 1
#define P_ROW_DONE(i, w_) cnt + DF_P + (w_), (i) * (N_OF(w_) + 1)

  if (wv < MAC_NP) {
    // ---- P: wave w works on strips w, w + MAC_NP, .. of every row ----
    const int w = wv, o1 = (w + 1) % MAC_NP, o2 = (w + 2) % MAC_NP, o3 = (w + 3) % MAC_NP;
    static_assert(MAC_NP >= 2 && MAC_NP <= 4, "the row-end wait names three waves (with fewer than four, some of them twice or this wave itself)");
    // active ranges of rows i .. i+3 (the last one requested a row before it is needed) and the strips of rows i, i+1
    int2 rA = rng_or_none(a, k, 1), rB = rng_or_none(a, k, 2), rC = rng_or_none(a, k, 3), rD = rC;
    StripSpan sp = span_fwd(sparse, rA, rB, ns), spn = sp;
    // !STAGE: the mask byte of this wave's NEXT unit (fetched behind the unit before it, or at the end of the row before)
    auto fetch_co = [&](int ni, int s_, int2 r) -> unsigned char {
      const int nj = 1 + (s_ << 6) + lane;
      return (ni <= Lq && nj <= Lt && rng_hits(r, 1 + (s_ << 6), 64 + (s_ << 6))) ? h.co[(size_t)ni * pitch + nj] : (unsigned char)1;
    };
    unsigned char co_next = 1;
    if (!STAGE && first_own(sp.sa, w, MAC_NP) <= sp.sb) co_next = fetch_co(1, first_own(sp.sa, w, MAC_NP), rA);
    double pmin = LOCAL ? 1.0 : 0.0, scale_prod = 1.0, scale_i = 1.0;
    // lanes 0-19: q.p[i], 20-24: q.tr[i-1][M2M, I2M, D2M, M2D, D2D], 25: q.tr[i][M2I]
    auto qrow = [&](int i) -> float {
      if (i > Lq || lane > 25) return 0.0f;
      if (lane < 20) return h.qp[(size_t)i * 20 + lane];
      const int tr = lane == 20 ? T_M2M : lane == 21 ? T_I2M : lane == 22 ? T_D2M : lane == 23 ? T_M2D : lane == 24 ? T_D2D : T_M2I;
      return h.qtr[(size_t)(lane == 25 ? i : i - 1) * 7 + tr];
    };
    float q_next = qrow(1);
    for (int i = 1; i <= Lq + 1; ++i) {
      const int cur = i & 1, prv = cur ^ 1;
      if (i >= 2) {
        // the end of row i-1 (both waves, the same operations): its rescaling factor needs the other wave's strips too
        DF_EVENT(7, i, 7)
        DF_WAIT(P_ROW_DONE(i - 1, o1), dead, P_ROW_DONE(i - 1, o2), P_ROW_DONE(i - 1, o3));
        double scale_next = 1.0;
        if (i - 1 >= 2) {
          double Pmax = pmaxring[prv * MAC_NP];
#pragma unroll
          for (int e = 1; e < MAC_NP; ++e) Pmax = fmax(Pmax, pmaxring[prv * MAC_NP + e]);
          pmin *= scale_i;
          if (pmin < DBL_MIN * 100) pmin = 0.0;
          scale_next = 1.0 / (Pmax + 1.0);  // :155
        }
        if (w == 0) {
          if (lane == 0) {
            if (i - 1 >= 2) h.scale[i] = scale_next;
            rring[prv] = scale_next;
          }
          df_post(cnt + DF_R, i - 1, lane);  // the total reads it at the end of its row
        }
        scale_i = scale_next;
        if (i > Lq) break;
        if (scale_prod < DBL_MIN * 100)
          scale_prod = 0.0;
        else
          scale_prod *= scale_i;
      }
      DF_EVENT(7, i, 8)
      if (i >= 2) {
        rA = rB, rB = rC, rC = rD;
        sp = spn;
      }
      rD = rng_or_none(a, k, i + 3);
      spn = span_fwd(sparse, rB, rC, ns);  // row i + 1
      // sparse rows: this row's buffer held row i-2, whose chains (the waves of this row's parity) may have units that no unit
      // of row i-1 waited for - they must be through before anything of row i is written (dense rows: implied by the unit waits)
      if (sparse && i >= 3) DF_WAIT(cnt + DF_S + 2 * cur + 0, ((i - 1) >> 1) * ns, dead, cnt + DF_S + 2 * cur + 1, ((i - 1) >> 1) * ns);
      const float q_cur = q_next;
      q_next = qrow(i + 1);
      float qi[20];
#pragma unroll
      for (int e = 0; e < 20; ++e) qi[e] = rl_f(q_cur, e);
      const double qM2M = rl_f(q_cur, 20), qI2M = rl_f(q_cur, 21), qD2M = rl_f(q_cur, 22), qM2D = rl_f(q_cur, 23), qD2D = rl_f(q_cur, 24);
      const double qM2I = rl_f(q_cur, 25);
      DF_EVENT(7, i, 9)
      double Pmax = 0.0;
      // the mask bytes of this wave's strips of the next row (entry q: strip w + q * MAC_NP), fetched now, parked in LDS at the
      // end of the row
      unsigned char pre_co[MAC_OWN];
      if (STAGE) {
#pragma unroll
        for (int q = 0; q < MAC_OWN; ++q) {
          const int u = w + q * MAC_NP, jn = 1 + u * 64 + lane;
          pre_co[q] = 1;
          if (u < ns && i < Lq && jn <= Lt) pre_co[q] = h.co[(size_t)(i + 1) * pitch + jn];
        }
      }
      DF_EVENT(8, i, 0)
      const unsigned char* co_row = sCo + cur * co_stride;
      const int above = ((i - 2) >> 1) * ns;  // units the sweep waves of row i-1's parity have finished before that row
      for (int s = first_own(sp.sa, w, MAC_NP); s <= sp.sb; s += MAC_NP) {
        if (i >= 2) {
          // row i-1's chains of this strip (and, in order, of the ones left of it) are final; its operands XB(.., strip s) and
          // its mask are consumed (the chain waves and the total are done with them), and so is everything of row i-2, whose
          // buffer this unit overwrites: both P waves finished row i-1 (above), and they waited for the readers of row i-2
          // before every unit of row i-1
          DF_WAIT(cnt + DF_S + 2 * (prv) + 0, above + s + 1, dead, cnt + DF_S + 2 * (prv) + 1, above + s + 1, cnt + DF_T, (i - 2) * ns + s + 1);
        }
        DF_EVENT(1, i, s)
        const int s0 = s << 6, j = 1 + s0 + lane;
        const bool valid = j <= Lt;
        const int jc = valid ? j : Lt;
        const bool off = !valid || (STAGE ? co_row[jc] != 0 : co_next != 0);
        if (!STAGE && s + MAC_NP <= sp.sb) co_next = fetch_co(i, s + MAC_NP, rA);  // this wave's next unit of the row
        const unsigned long long on_mask = __ballot(!off);
        if (lane == 0) masks[cur * MAC_DF_STRIPS + MSK(s)] = on_mask;
        if (on_mask == 0) {
          // a strip without a single active cell: all five states are zero, the running sums are unchanged
          if (valid) {
            ROW(cur, F_MM, j) = 0.0;
            ROW(cur, F_GD, j) = 0.0;
            ROW(cur, F_IM, j) = 0.0;
            ROW(cur, F_DG, j) = 0.0;
            ROW(cur, F_MI, j) = 0.0;
            XB(2, j) = 0.0;
          }
        } else {
          float tpj[20], tt1[7], tt[7];
          load_tp<STAGE>(h, sTp, jc, tpj);
          load_tt<STAGE>(h, sTt, jc - 1, tt1);  // t.tr[j-1]
          load_tt<STAGE>(h, sTt, jc, tt);       // t.tr[j]
          const float pf = dot20(qi, tpj);
          // fpow2(ScoreSS(q, t, i, j)); for column 1 the reference passes (1, j) with the stale loop variable j = t.L + 1 (:77)
          float ssf = 1.0f;
          if (h.ssm && i >= 2) ssf = j == 1 ? sstab[h.ssq[1] * h.sstw + h.sst[Lt + 1]] : sstab[h.ssq[i] * h.sstw + h.sst[jc]];
          double mm, dg, mi;
          if (i == 1) {
            mm = pf * Cshift;  // :31
            dg = mi = 0.0;
          } else {
            const double pm = ROW(prv, F_MM, jc), pdg = ROW(prv, F_DG, jc), pmi = ROW(prv, F_MI, jc);
            if (j == 1) {
              mm = scale_prod * ssf * pf * Cshift;  // :71-73
            } else {
              const double m1 = ROW(prv, F_MM, jc - 1), g1 = ROW(prv, F_GD, jc - 1), i1 = ROW(prv, F_IM, jc - 1),
                           d1 = ROW(prv, F_DG, jc - 1), x1 = ROW(prv, F_MI, jc - 1);
              mm = pf * Cshift * ssf * scale_i *
                   (pmin + m1 * qM2M * tt1[T_M2M] + g1 * qM2M * tt1[T_D2M] + i1 * qI2M * tt1[T_M2M] + d1 * qD2M * tt1[T_M2M] +
                    x1 * qM2M * tt1[T_I2M]);  // :94-103
            }
            dg = scale_i * (pm * qM2D + pdg * qD2D);                          // :110-112 / :79-81
            mi = scale_i * (pm * qM2M * tt[T_M2I] + pmi * qM2M * tt[T_I2I]);  // :113-116 / :75-78
          }
          if (off) mm = dg = mi = 0.0;
          if (i >= 2 && j >= 2 && !off) Pmax = fmax(Pmax, mm);
          if (valid) {
            ROW(cur, F_MM, j) = mm;
            ROW(cur, F_DG, j) = dg;
            ROW(cur, F_MI, j) = mi;
            XB(2, j) = (double)(float)mm;  // what p_mm stores (float) and Pforward sums (:168): the total's wave stores it
          }
          // the recurrences along the row (:104-109): gd = mm(j-1)*t[j-1][M2D] + gd(j-1)*t[j-1][D2D],
          //                                           im = mm(j-1)*q[i][M2I]*t[j-1][M2M] + im(j-1)*q[i][I2I]*t[j-1][M2M]
          // mm(j-1) of the strip's first column is the other wave's (the strip to the left)
          double left = 0.0;
          if (s > sp.sa) {  // (the strip left of the row's first one holds no active cell: its F_MM is 0)
            DF_WAIT(P_DONE(i, s - 1), dead);
            left = ROW(cur, F_MM, s0);
          }
          const double mm_left = shr1_d(mm, left);
          const bool chain_on = !off && (i == 1 || j >= 2);  // column 1 of rows >= 2: im = gd = 0 (:74)
          const double a_gd = chain_on ? mm_left * tt1[T_M2D] : 0.0, b_gd = chain_on ? (double)tt1[T_D2D] : 0.0;
          const double c_im = chain_on ? mm_left * qM2I * tt1[T_M2M] : 0.0, b_im = chain_on ? (double)tt1[T_M2M] : 0.0;
          if (valid) {
            ROW(cur, F_GD, j) = a_gd;
            ROW(cur, F_IM, j) = c_im;
            XB(0, j) = b_gd;
            XB(1, j) = b_im;
          }
        }
#if defined(HHV_EXP_MAC_TIMEOUT)
        if (w != 0)
#endif
        df_post(cnt + DF_P + w, (i - 1) * (N_OF(w) + 1) + s / MAC_NP + 1, lane);
        DF_EVENT(2, i, s)
      }
      // the end of this wave's row: the next row's mask bytes, the row's maximum, the row's last unit number
      if (STAGE) {
        unsigned char* co_nextrow = sCo + prv * co_stride;
#pragma unroll
        for (int q = 0; q < MAC_OWN; ++q) {
          const int u = w + q * MAC_NP, jn = 1 + u * 64 + lane;
          if (u < ns && jn <= Lt) co_nextrow[jn] = pre_co[q];
        }
      } else if (i < Lq && first_own(spn.sa, w, MAC_NP) <= spn.sb) {
        co_next = fetch_co(i + 1, first_own(spn.sa, w, MAC_NP), rB);
      }
      Pmax = wave_max_d(Pmax);
      if (lane == 0) pmaxring[cur * MAC_NP + w] = Pmax;
#if defined(HHV_EXP_MAC_TIMEOUT)
      if (w != 0)
#endif
      df_post(cnt + DF_P + w, i * (N_OF(w) + 1), lane);
    }
  } else if (wv < MAC_NP + 4) {
    // ---- the GD / IM chains of the rows of one parity ----
    const int par = (wv - MAC_NP) >> 1;
    const bool gdw = ((wv - MAC_NP) & 1) == 0;
    const LdsCnt mine = cnt + DF_S + 2 * par + (gdw ? 0 : 1);
    float qI2I_next = (!gdw && (par ? 1 : 2) <= Lq) ? h.qtr[(size_t)(par ? 1 : 2) * 7 + T_I2I] : 0.0f;  // fetched a row ahead
    int2 rA = rng_or_none(a, k, par ? 1 : 2), rB = rng_or_none(a, k, par ? 2 : 3);  // ranges of rows i, i + 1 (a row of this wave ahead)
    for (int i = par ? 1 : 2; i <= Lq; i += 2) {
      const int cur = i & 1;
      const double qI2I = qI2I_next;
      qI2I_next = (!gdw && i + 2 <= Lq) ? h.qtr[(size_t)(i + 2) * 7 + T_I2I] : 0.0f;
      const StripSpan sp = span_fwd(sparse, rA, rB, ns);
      rA = rng_or_none(a, k, i + 2), rB = rng_or_none(a, k, i + 3);
      const int base = ((i - 1) >> 1) * ns;  // units of this wave before row i
      double carry = 0.0;
      for (int s = sp.sa; s <= sp.sb; ++s) {
        DF_WAIT(P_DONE(i, s), dead);
        DF_EVENT(3, i, s)
        const unsigned long long on_mask = masks[cur * MAC_DF_STRIPS + MSK(s)];
        if (on_mask == 0) {
          carry = 0.0;
        } else {
          // lane 0 walks the active span l0 .. l1; what enters it is the carry of the previous strip when the span starts at the
          // strip's first column, else 0 (an inactive neighbour); inactive columns inside the span hold c = b = 0 (y becomes 0,
          // as in the sweep), the columns outside it hold c = 0, which is their result
          const int l0 = __builtin_ctzll(on_mask), l1 = 63 - __builtin_clzll(on_mask);
          const int jf = 1 + (s << 6) + l0, n = l1 - l0 + 1;
          if (lane == 0) {
            const double yin = l0 == 0 ? carry : 0.0;
            const double y = gdw ? mac_walk<1, 0>(&ROW(cur, F_GD, jf), &XB(0, jf), n, yin, 0.0)
                                 : mac_walk<1, 1>(&ROW(cur, F_IM, jf), &XB(1, jf), n, yin, qI2I);
            carry = l1 == 63 ? y : 0.0;
          }
        }
        df_post(mine, base + s + 1, lane);
        DF_EVENT(4, i, s)
      }
      df_post(mine, base + ns, lane);
    }
  } else {
    // ---- the total forward probability (:162-182), one chain through all units ----
    double Pf = LOCAL ? 1.0 : 0.0;
    int2 rA = rng_or_none(a, k, 1), rB = rng_or_none(a, k, 2);
    StripSpan sp_last = span_fwd(sparse, rA, rB, ns);
    for (int i = 1; i <= Lq; ++i) {
      const int cur = i & 1;
      const StripSpan sp = span_fwd(sparse, rA, rB, ns);
      sp_last = sp;
      rA = rB, rB = rng_or_none(a, k, i + 2);
      if (lane == 0) h.mat[(size_t)i * pitch] = 0.0f;
      for (int s = sp.sa; s <= sp.sb; ++s) {
        DF_WAIT(P_DONE(i, s), dead);
        DF_EVENT(5, i, s)
        {
          // F_MM as the float the reference keeps (p_mm): stored by this wave, which never waits for global memory
          const int j = 1 + (s << 6) + lane;
          if (j <= Lt) h.mat[(size_t)i * pitch + j] = (float)XB(2, j);
        }
        if (LOCAL) {
          const unsigned long long on_mask = masks[cur * MAC_DF_STRIPS + MSK(s)];
          if (on_mask != 0) {
            // the summands of the active span in column order (inactive columns hold 0)
            const int l0 = __builtin_ctzll(on_mask), l1 = 63 - __builtin_clzll(on_mask);
            if (lane == 0) Pf = mac_walk<1, 2>(&XB(2, 1 + (s << 6) + l0), nullptr, l1 - l0 + 1, Pf, 0.0);
          }
        }
        df_post(cnt + DF_T, (i - 1) * ns + s + 1, lane);
        DF_EVENT(6, i, s)
      }
      {
        // the end of the row (a sparse row's last strip need not be the template's: F_MM of column Lt is then 0)
        DF_WAIT(cnt + DF_R, i, dead);
        const double scale_next = rring[cur];
        if (LOCAL) {
          Pf *= scale_next;
        } else if (i < Lq) {
          const double fL = sp.sa <= sp.sb && sp.sb == ns - 1 ? ROW(cur, F_MM, Lt) : 0.0;
          Pf = (Pf + (float)fL * scale_next);
        }
        df_post(cnt + DF_T, i * ns, lane);
      }
    }
    Pf = lane_d(Pf, 0);  // (lane 0 carried it)
    if (!LOCAL) {
      // + sum_j F(Lq, j) in column order, then * scale[Lq+1]
      const int last = Lq & 1;
      for (int s0 = 0; s0 < Lt; s0 += 64) {
        const int j = 1 + s0 + lane;
        const bool seen = (s0 >> 6) >= sp_last.sa && (s0 >> 6) <= sp_last.sb;  // (strips row Lq did not visit hold an older row)
        const double f_mm = (j <= Lt && seen) ? (double)(float)ROW(last, F_MM, j) : 0.0;
        double acc = Pf;
        for (int q = 0; q < 64; ++q) acc = shr1_d(acc, Pf) + f_mm;
        Pf = lane_d(acc, 63);
      }
      Pf *= rring[Lq & 1];
    }
    if (lane == 0) a.Pforward[k] = Pf;
  }
  df_report(dead, a.err, lane);
  DF_TIMING_REPORT("forward")
#undef P_DONE
#undef P_ROW_DONE
}
#undef DFC

#undef MAC_NP
#define MAC_NP MAC_NP_BWD
// Backward: P (wave 0) computes what depends on row i+1 only - pmatch, DG, MI, the chains' operands and, for B_MM, the partial
// sum pmin + pmatch*q[M2M]*t[M2M] (into the F_MM slot) and the two last summands (XB(2 / 3, j)); the sweep waves follow; P2
// (wave 5) completes B_MM = (((partial + gd(j+1)*t[M2D]) + im(j+1)*q[M2I]*t[M2M]) + XB2) + XB3 - the reference's left-to-right
// sum (src/hhbackwardalgorithm.cpp:86-93) - and turns F_MM into the posterior.  P of (i-1, s) waits for P2 of (i, s).
#define DFC(j) mac_col_b<RING>((j), Lt)
template <bool LOCAL, bool STAGE, bool LISTS, bool RING = false>
__global__ void __launch_bounds__(MAC_DFB_THREADS) hhv_mac_backward_df_kernel(MacArgs a) {
  static_assert(!(RING && STAGE), "the ring is for templates that do not fit LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* rows = reinterpret_cast<double*>(smem);
  constexpr int NT = MAC_DFB_THREADS;
  const int k = a.sel[blockIdx.x], tid = threadIdx.x, lane = tid & 63;
  if (RING && (a.Lt[k] < a.ring_min_Lt || (a.row_rng + (size_t)k * (a.Lq + 2))[0].x > MAC_RING_STRIPS)) return;  // a short template in this class for capacity, or a row wider than the ring: the single-wave kernels
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  DF_TIMING_DECL
  const HitView h = view(a, k);
  const int Lq = h.Lq, Lt = h.Lt, pitch = h.pitch, stride = RING ? MAC_RING_COLS + 2 : Lt + 2;
  const size_t cols = (size_t)a.lds_cols + 2;
  double* ctl = rows + MAC_ROW_FIELDS * cols;
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(ctl);  // [2][MAC_DF_STRIPS]
  double* prow = ctl + 2 * MAC_DF_STRIPS;  // [2][4] what the posterior waves need of row i: mask byte of (i, Lt) != 0, scale[i+1], q.tr[i][M2I]
  int* cnt_mem = reinterpret_cast<int*>(prow + 8);
  const LdsCnt cnt = lds_cnt(cnt_mem);
  float* sTp = reinterpret_cast<float*>(ctl + MAC_CTL_DOUBLES);
  float* sTt = sTp + cols * 20;
  // STAGE: the mask bytes (P) and F_MM (P2) of a row are fetched while the row processed before it is computed
  unsigned char* sCo = reinterpret_cast<unsigned char*>(sTt + cols * 8);  // [2][cols]
  const int co_stride = (int)cols;
  float* sF = reinterpret_cast<float*>(sCo + (((size_t)2 * co_stride + 15) & ~(size_t)15));  // [2][cols]
  float* sSs = sF + (size_t)2 * co_stride;                                                    // [352]
  const float* sstab = STAGE ? sSs : h.sstab;
  const double Cshift = a.Cshift, Pf = a.Pforward[k];
  for (int e = tid; e < 10 * stride; e += NT) rows[e] = 0.0;
  if (STAGE) {
    stage_template_wg(h, sTp, sTt, tid, NT);
    if (h.ssm)
      for (int e = tid; e < 352; e += NT) sSs[e] = h.sstab[e];
    if (Lq >= 2)
      for (int j = 1 + tid; j <= Lt; j += NT) {  // row Lq - 1, the first one of the loops below
        sCo[((Lq - 1) & 1) * co_stride + j] = h.co[(size_t)(Lq - 1) * pitch + j];
        sF[((Lq - 1) & 1) * co_stride + j] = h.mat[(size_t)(Lq - 1) * pitch + j];
      }
  }
  if (tid < DF_N) cnt_mem[tid] = 0;
  __syncthreads();
  const double sL = h.scale[Lq + 1];
  const int ns = Lt >= 2 ? (Lt - 1 + 63) >> 6 : 1;  // strips of columns Lt-1 .. 1 (Lt = 1: one strip without a valid lane)
  // row Lq (:19-29); row i lives in buffer i & 1.  (RING: only the strips row Lq-1 reads - and column Lt - have a slot)
  {
    const StripSpan spq = span_bwd(true, rng_or_none(a, k, Lq), rng_or_none(a, k, Lq - 1), ns, Lt);
    for (int j = 1 + tid; j <= Lt; j += NT) {
      float* pv = h.mat + (size_t)Lq * pitch + j;
      const int sj = (Lt - 1 - j) >> 6;
      const bool slot = !RING || j == Lt || (sj >= spq.sa && sj <= spq.sb);
      if (h.co[(size_t)Lq * pitch + j]) {
        *pv = 0.0f;
        if (slot) ROW(Lq & 1, F_MM, j) = 0.0;
      } else {
        if (slot) ROW(Lq & 1, F_MM, j) = sL;
        *pv = (float)(*pv * sL / Pf);
      }
    }
  }
  __syncthreads();
  // sparse rows (StripSpan); a hit whose Pforward is not a positive number keeps every strip: the reference's posterior is then
  // F * (float)(B / Pforward) = NaN in EVERY cell, which only a visit writes
  // (RING cannot keep every strip: its posterior waves fill the cells no unit visits with the value a visit would write)
  const bool pf_ok = Pf > 0.0 && Pf < 1.0e300;
  const bool sparse = RING || (Lt >= a.sparse_min_Lt && pf_ok);
  const LdsCnt dead = cnt + DF_DEAD;
  // unit (i, s) of the parallel part is done when its wave's counter has reached ...
#define P_DONE(i, s) cnt + DF_P + ((s) % MAC_NP), (Lq - 1 - (i)) * N_OF((s) % MAC_NP) + (s) / MAC_NP + 1
  // ... and the posterior wave's (even / odd strips)
#define T_DONE(i, s) cnt + DF_T + ((s)&1), (Lq - 1 - (i)) * ((ns + 1 - ((s)&1)) >> 1) + ((s) >> 1) + 1

  if (wv < MAC_NP) {
    // ---- P: wave w works on strips w, w + MAC_NP, .. of every row (the units of a row do not depend on each other) ----
    const int w = wv;
    double pmin = LOCAL ? sL : 0.0;
    double sc_next = Lq >= 2 ? h.scale[Lq] : 1.0;  // scale[i+1] of the row, fetched a row ahead
    // active ranges of rows i, i-1, i-2, i-3 (the last one requested a row before it is needed), strips of rows i and i-1
    int2 rA = rng_or_none(a, k, Lq - 1), rB = rng_or_none(a, k, Lq - 2), rC = rng_or_none(a, k, Lq - 3), rD = rC;
    StripSpan sp = span_bwd(sparse, rA, rB, ns, Lt), spn = sp;
    // !STAGE: the mask byte of this wave's NEXT unit
    auto fetch_co = [&](int ni, int s_, int2 r) -> unsigned char {
      const int nj = Lt - 1 - (s_ << 6) - lane;
      return (ni >= 1 && nj >= 1 && rng_hits(r, Lt - 64 - (s_ << 6), Lt - 1 - (s_ << 6))) ? h.co[(size_t)ni * pitch + nj] : (unsigned char)1;
    };
    unsigned char co_nx = 1;
    if (!STAGE && Lq >= 2 && first_own(sp.sa, w, MAC_NP) <= sp.sb) co_nx = fetch_co(Lq - 1, first_own(sp.sa, w, MAC_NP), rA);
    // lanes 0-19: q.p[i+1], 20-25: q.tr[i][M2M, M2D, I2M, D2M, D2D, M2I]
    auto qrow = [&](int i) -> float {
      if (i < 1 || lane > 25) return 0.0f;
      if (lane < 20) return h.qp[(size_t)(i + 1) * 20 + lane];
      const int tr = lane == 20 ? T_M2M : lane == 21 ? T_M2D : lane == 22 ? T_I2M : lane == 23 ? T_D2M : lane == 24 ? T_D2D : T_M2I;
      return h.qtr[(size_t)i * 7 + tr];
    };
    float q_next = qrow(Lq - 1);
    for (int i = Lq - 1; i >= 1; --i) {
      const int cur = i & 1, prv = cur ^ 1;
      const double sc = sc_next;
      sc_next = i >= 2 ? h.scale[i] : 1.0;
      pmin *= sc;
      if (pmin < DBL_MIN * 100) pmin = 0.0;
      if (i <= Lq - 2) {
        rA = rB, rB = rC, rC = rD;
        sp = spn;
      }
      rD = rng_or_none(a, k, i - 3);
      spn = span_bwd(sparse, rB, rC, ns, Lt);  // row i - 1
      const float q_cur = q_next;
      q_next = qrow(i - 1);
      float qn[20];
#pragma unroll
      for (int e = 0; e < 20; ++e) qn[e] = rl_f(q_cur, e);
      const double qM2M = rl_f(q_cur, 20), qM2D = rl_f(q_cur, 21), qI2M = rl_f(q_cur, 22), qD2M = rl_f(q_cur, 23), qD2D = rl_f(q_cur, 24);
      const unsigned char* corow = h.co + (size_t)i * pitch;
      const unsigned char* co_l = sCo + cur * co_stride;
      // the mask bytes and the F_MM values of this wave's strips of the next row (entry q: strip w + q * MAC_NP; wave 0 lane 0 also
      // column Lt): fetched now, parked in LDS at the end of the row.  Every entry has one writer, and the posterior waves, which
      // store to global memory all the time, never have to wait for a load.
      unsigned char pre_co[MAC_OWN], pre_coL = 1;
      float pre_f[MAC_OWN], pre_fL = 0.0f;
      if (STAGE) {
#pragma unroll
        for (int q = 0; q < MAC_OWN; ++q) {
          const int u = w + q * MAC_NP, jn = Lt - 1 - (u << 6) - lane;
          pre_co[q] = 1, pre_f[q] = 0.0f;
          if (u < ns && i >= 2 && jn >= 1) pre_co[q] = h.co[(size_t)(i - 1) * pitch + jn], pre_f[q] = h.mat[(size_t)(i - 1) * pitch + jn];
        }
        if (w == 0 && lane == 0 && i >= 2) pre_coL = h.co[(size_t)(i - 1) * pitch + Lt], pre_fL = h.mat[(size_t)(i - 1) * pitch + Lt];
      }
      // (!STAGE: column Lt outside the row's active range is masked - no trip to memory)
      const unsigned char coL = STAGE ? co_l[Lt] : ((rA.x <= rA.y && Lt <= rA.y) ? corow[Lt] : (unsigned char)1);  // the mask byte of column Lt, for P2's column-Lt step of this row
      if (w == 0) {
        // the row's constants for the posterior waves.  The slot held row i+2's: both posterior waves are through with that row
        // when they have started on row i+1, which the units of row i wait for - a wave without a unit in this (sparse) row waits here
        if (i <= Lq - 3) DF_WAIT(cnt + DF_T + 0, (Lq - 2 - i) * ((ns + 1) >> 1), dead, cnt + DF_T + 1, (Lq - 2 - i) * (ns >> 1));
        if (lane == 0) {
          prow[cur * 4 + 0] = coL ? 1.0 : 0.0;
          prow[cur * 4 + 1] = sc;
          prow[cur * 4 + 2] = rl_f(q_cur, 25);
        }
        df_post(cnt + DF_R, Lq - i, lane);  // prow of row i is there (rows count from Lq - 1 = 1)
      }
      for (int s = first_own(sp.sa, w, MAC_NP); s <= sp.sb; s += MAC_NP) {
        // B_MM of row i+1 at this strip's columns and the one right of them is final; with it the chain waves of (i+1, s) are
        // done with XB(0 / 1), P2 with XB(2 / 3), the mask and the per-row ring.  Row i+2, whose buffer this unit overwrites,
        // has been read by this wave's units of row i+1 and by P2 (this wave waited for P2 of (i+2, s) before (i+1, s)); the
        // other P wave's unit (i+1, s+1) reads one column of strip s of row i+2: wait for it too
        if (i <= Lq - 2) {
          // (B_MM of the column right of the strip is the other posterior wave's, unit (i+1, s-1))
          if (s > 0) {
            DF_WAIT(T_DONE(i + 1, s), dead, T_DONE(i + 1, s - 1));
          } else {
            DF_WAIT(T_DONE(i + 1, s), dead);
          }
          if (s + 1 < ns) DF_WAIT(P_DONE(i + 1, s + 1), dead);
        }
        DF_EVENT(1, i, s)
        const int j = Lt - 1 - (s << 6) - lane;  // descending: lane 0 is the rightmost column of the strip
        const bool valid = j >= 1;
        const int jc = valid ? j : 1;
        const bool off = !valid || (STAGE ? co_l[jc] != 0 : co_nx != 0);
        if (!STAGE && s + MAC_NP <= sp.sb) co_nx = fetch_co(i, s + MAC_NP, rA);  // this wave's next unit of the row
        const unsigned long long on_mask = __ballot(!off);
        if (lane == 0) masks[cur * MAC_DF_STRIPS + MSK(s)] = on_mask;
        if (on_mask == 0) {
          if (valid) {
            ROW(cur, F_MM, j) = 0.0;
            ROW(cur, F_GD, j) = 0.0;
            ROW(cur, F_IM, j) = 0.0;
            ROW(cur, F_DG, j) = 0.0;
            ROW(cur, F_MI, j) = 0.0;
          }
        } else {
          float tpn[20], tt[7];
          load_tp<STAGE>(h, sTp, jc + 1, tpn);
          load_tt<STAGE>(h, sTt, jc, tt);
          const float pf = dot20(qn, tpn);
          const float ssf = h.ssm ? sstab[h.ssq[i + 1] * h.sstw + h.sst[jc + 1]] : 1.0f;  // fpow2(ScoreSS(q, t, i+1, j+1))
          const double pmatch = ROW(prv, F_MM, jc + 1) * pf * ssf * Cshift * sc;  // :80-83
          const double pdg = ROW(prv, F_DG, jc), pmi = ROW(prv, F_MI, jc);
          const double tM2M = tt[T_M2M];
          double dg = (+pmatch * qD2M * tM2M + pdg * qD2D * sc);                    // :103-106
          double mi = (+pmatch * qM2M * tt[T_I2M] + pmi * qM2M * tt[T_I2I] * sc);  // :108-111
          const double a_gd = off ? 0.0 : pmatch * qM2M * tt[T_D2M], b_gd = off ? 0.0 : (double)tt[T_D2D];  // :95-97
          const double c_im = off ? 0.0 : pmatch * qI2M * tM2M, b_im = off ? 0.0 : tM2M;                    // :99-101
          const double t0 = (+pmin + pmatch * qM2M * tM2M);  // :86-93, the first two summands
          const double e4 = pdg * qM2D * sc, e5 = pmi * qM2M * tt[T_M2I] * sc;
          if (off) dg = mi = 0.0;
          if (valid) {
            ROW(cur, F_MM, j) = t0;
            ROW(cur, F_GD, j) = a_gd;
            ROW(cur, F_IM, j) = c_im;
            ROW(cur, F_DG, j) = dg;
            ROW(cur, F_MI, j) = mi;
            XB(0, j) = b_gd;
            XB(1, j) = b_im;
            XB(2, j) = e4;
            XB(3, j) = e5;
          }
        }
#if defined(HHV_EXP_MAC_TIMEOUT)
        if (w != 0)
#endif
        df_post(cnt + DF_P + w, (Lq - 1 - i) * N_OF(w) + s / MAC_NP + 1, lane);
        DF_EVENT(2, i, s)
      }
      // the end of this wave's row
      if (STAGE) {
        // the next row's mask bytes and F_MM values replace row i+1's, which the posterior waves read: the units of row i waited
        // for them strip by strip; a sparse row has not visited every strip, so wait for the end of their row i+1
        if (sparse && i <= Lq - 2) DF_WAIT(cnt + DF_T + 0, (Lq - 1 - i) * ((ns + 1) >> 1), dead, cnt + DF_T + 1, (Lq - 1 - i) * (ns >> 1));
        unsigned char* co_n = sCo + prv * co_stride;
        float* f_n = sF + prv * co_stride;
#pragma unroll
        for (int q = 0; q < MAC_OWN; ++q) {
          const int u = w + q * MAC_NP, jn = Lt - 1 - (u << 6) - lane;
          if (u < ns && jn >= 1) co_n[jn] = pre_co[q], f_n[jn] = pre_f[q];
        }
        if (w == 0 && lane == 0) co_n[Lt] = pre_coL, f_n[Lt] = pre_fL;
      } else if (i >= 2 && first_own(spn.sa, w, MAC_NP) <= spn.sb) {
        co_nx = fetch_co(i - 1, first_own(spn.sa, w, MAC_NP), rB);
      }
#if defined(HHV_EXP_MAC_TIMEOUT)
      if (w != 0)
#endif
      df_post(cnt + DF_P + w, (Lq - i) * N_OF(w), lane);
    }
  } else if (wv < MAC_NP + 4) {
    // ---- the GD / IM chains of the rows of one parity ----
    const int par = (wv - MAC_NP) >> 1;
    const bool gdw = ((wv - MAC_NP) & 1) == 0;
    const LdsCnt mine = cnt + DF_S + 2 * par + (gdw ? 0 : 1);
    const int i_first = ((Lq - 1) & 1) == par ? Lq - 1 : Lq - 2;
    float qI2I_next = (!gdw && i_first >= 1) ? h.qtr[(size_t)i_first * 7 + T_I2I] : 0.0f;  // fetched a row ahead
    int2 rA = rng_or_none(a, k, i_first), rB = rng_or_none(a, k, i_first - 1);  // ranges of rows i, i - 1 (a row of this wave ahead)
    for (int i = i_first; i >= 1; i -= 2) {
      const int cur = i & 1;
      const double qI2I = qI2I_next;
      qI2I_next = (!gdw && i - 2 >= 1) ? h.qtr[(size_t)(i - 2) * 7 + T_I2I] : 0.0f;
      const StripSpan sp = span_bwd(sparse, rA, rB, ns, Lt);
      rA = rng_or_none(a, k, i - 2), rB = rng_or_none(a, k, i - 3);
      const int base = ((Lq - 1 - i) >> 1) * ns;  // units of this wave before row i
      double carry = 0.0;  // curr[Lt].gd = curr[Lt].im = 0
      for (int s = sp.sa; s <= sp.sb; ++s) {
        DF_WAIT(P_DONE(i, s), dead);
        DF_EVENT(3, i, s)
        const unsigned long long on_mask = masks[cur * MAC_DF_STRIPS + MSK(s)];
        if (on_mask == 0) {
          carry = 0.0;
        } else {
          // as in the forward kernel, towards smaller columns: lane l0 is the span's rightmost column
          const int l0 = __builtin_ctzll(on_mask), l1 = 63 - __builtin_clzll(on_mask);
          const int jf = Lt - 1 - (s << 6) - l0, n = l1 - l0 + 1;
          if (lane == 0) {
            const double yin = l0 == 0 ? carry : 0.0;
            const double y = gdw ? mac_walk<-1, 0>(&ROW(cur, F_GD, jf), &XB(0, jf), n, yin, 0.0)
                                 : mac_walk<-1, 1>(&ROW(cur, F_IM, jf), &XB(1, jf), n, yin, qI2I);
            carry = l1 == 63 ? y : 0.0;
          }
        }
        df_post(mine, base + s + 1, lane);
        DF_EVENT(4, i, s)
      }
      df_post(mine, base + ns, lane);
    }
  } else {
    // ---- P2: wave v works on strips v, v+2, .. of every row ----
    const int v = wv - (MAC_NP + 4);
    double scale_prod = sL;
    double final_scale_prod = sL;  // :31-36
    if (LISTS) {
      // (64 factors per trip to memory: the other waves wait for this one's first unit)
      for (int base = Lq - 1; base >= 1; base -= 64) {
        const int ii = base - lane;
        const double sv = ii >= 1 ? h.scale[ii + 1] : 1.0;
        const int n = base < 64 ? base : 64;
        for (int l = 0; l < n; ++l) {
          final_scale_prod *= lane_d(sv, l);
          if (final_scale_prod < DBL_MIN * 100) final_scale_prod = 0.0;
        }
      }
    }
    float* blist = LISTS ? a.bwd_list + a.mat_off[k] : nullptr;
    // (!STAGE) F_MM of this wave's NEXT unit; a strip outside its row's active range holds the 0 the forward pass left
    int2 rA = rng_or_none(a, k, Lq - 1), rB = rng_or_none(a, k, Lq - 2), rC = rng_or_none(a, k, Lq - 3), rD = rC;
    StripSpan sp = span_bwd(sparse, rA, rB, ns, Lt), spn = sp;
    auto fetch_f = [&](int ni, int s_, int2 r) -> float {
      const int nj = Lt - 1 - (s_ << 6) - lane;
      return (ni >= 1 && nj >= 1 && rng_hits(r, Lt - 64 - (s_ << 6), Lt - 1 - (s_ << 6))) ? h.mat[(size_t)ni * pitch + nj] : 0.0f;
    };
    float f_nx = 0.0f;
    if (!STAGE && Lq >= 2 && first_own(sp.sa, v, 2) <= sp.sb) f_nx = fetch_f(Lq - 1, first_own(sp.sa, v, 2), rA);
    const int n_own = (ns + 1 - v) >> 1;  // strips of a row this wave owns
    // (every row passes here, with or without units of this wave: scale_prod sees every row's factor)
    for (int i = Lq - 1; i >= 1; --i) {
      const int cur = i & 1;
      if (i <= Lq - 2) {
        rA = rB, rB = rC, rC = rD;
        sp = spn;
      }
      rD = rng_or_none(a, k, i - 3);
      spn = span_bwd(sparse, rB, rC, ns, Lt);  // row i - 1
      float* row = h.mat + (size_t)i * pitch;
      const float* f_l = sF + cur * co_stride;  // STAGE: F_MM of the row, staged by the P waves
      // the row's constants, left by the first P wave before its units of the row
      DF_WAIT(cnt + DF_R, Lq - i, dead);
      {
        const double sc = prow[cur * 4 + 1];
        scale_prod *= sc;
        if (scale_prod < DBL_MIN * 100) scale_prod = 0.0;
      }
      const double qM2I = prow[cur * 4 + 2];
      if (v == 0 && lane == 0) {
        // column Lt (:58-71), ahead of every unit of the row (the row below reads it with its rightmost strip)
        const float fL = STAGE ? f_l[Lt] : ((rA.x <= rA.y && Lt <= rA.y) ? row[Lt] : 0.0f);
        if (prow[cur * 4 + 0] != 0.0) {
          row[Lt] = 0.0f;
          ROW(cur, F_MM, Lt) = 0.0;
        } else {
          ROW(cur, F_MM, Lt) = scale_prod;
          row[Lt] = (float)(fL * scale_prod / Pf);
        }
        ROW(cur, F_GD, Lt) = ROW(cur, F_IM, Lt) = ROW(cur, F_DG, Lt) = ROW(cur, F_MI, Lt) = 0.0;
      }
      for (int s = first_own(sp.sa, v, 2); s <= sp.sb; s += 2) {
        const int need = ((Lq - 1 - i) >> 1) * ns + s + 1;  // units of the chain waves of this row's parity
        DF_WAIT(cnt + DF_S + 2 * cur + 0, need, dead, cnt + DF_S + 2 * cur + 1, need);
        DF_EVENT(5, i, s)
        const int j = Lt - 1 - (s << 6) - lane;
        const bool valid = j >= 1;
        const int jc = valid ? j : 1;
        const float f_cur = STAGE ? (valid ? f_l[jc] : 0.0f) : f_nx;
        if (!STAGE && s + 2 <= sp.sb) f_nx = fetch_f(i, s + 2, rA);
        const unsigned long long on_mask = masks[cur * MAC_DF_STRIPS + MSK(s)];
        const bool off = !((on_mask >> lane) & 1);
        double mm = 0.0;
        if (on_mask != 0) {
          float tt[7];
          load_tt<STAGE>(h, sTt, jc, tt);
          const double tM2M = tt[T_M2M];
          double gr = ROW(cur, F_GD, jc + 1), ir = ROW(cur, F_IM, jc + 1);  // curr[j+1].gd / .im
          // (the strip right of a sparse row's first one holds no active cell - its chains are 0 - but was not visited)
          if (lane == 0 && s == sp.sa && s > 0) gr = ir = 0.0;
          mm = (ROW(cur, F_MM, jc) + gr * tt[T_M2D] + ir * qM2I * tM2M + XB(2, jc) + XB(3, jc));  // :86-93
          if (off) mm = 0.0;
          if (valid) ROW(cur, F_MM, j) = mm;
        }
        // B_MM is what the row below waits for: posted before the posterior is worked out
        df_post(cnt + DF_T + v, (Lq - 1 - i) * n_own + (s >> 1) + 1, lane);
        DF_EVENT(6, i, s)
        if (on_mask == 0) {
          if (valid) row[j] = f_cur * (float)(0.0 / Pf);  // F * (float)(B / Pforward) with B = 0 (NaN if Pforward is 0, as in the reference)
        } else {
          if (valid) row[j] = f_cur * (float)(mm / Pf);  // multiplyPosteriorValue(i, jj, float) (:122-124)
          if (LISTS && valid && !off) {
            // :112-122: float = ProbFwd(q.p[i], t.p[j]) * Cshift * B_MM / Pforward * final_scale_prod / scale_prod, left to right
            float tpj[20];
            load_tp<STAGE>(h, sTp, jc, tpj);
            const float sub = dot20(h.qp + (size_t)i * 20, tpj);
            const float lv = (float)((double)sub * Cshift * mm / Pf * final_scale_prod / scale_prod);
            if (lv > MAC_LIST_THRESHOLD) blist[(size_t)i * pitch + j] = lv;
          }
        }
      }
      if (RING && !pf_ok && v == 0) {
        // F * (float)(B / Pforward) of a cell nobody visited: F = 0 (cleared plane), B = 0
        const float fill = 0.0f * (float)(0.0 / Pf);
        const int c_hi = sp.sa <= sp.sb ? Lt - 1 - (sp.sa << 6) : 0, c_lo = sp.sa <= sp.sb ? Lt - 64 - (sp.sb << 6) : 1;  // visited columns c_lo .. c_hi
        for (int j = 1 + lane; j <= Lt - 1; j += 64)
          if (j < c_lo || j > c_hi) row[j] = fill;
      }
      if (!STAGE && i >= 2 && first_own(spn.sa, v, 2) <= spn.sb) f_nx = fetch_f(i - 1, first_own(spn.sa, v, 2), rB);
      df_post(cnt + DF_T + v, (Lq - i) * n_own, lane);  // the row's last unit number
    }
  }
  df_report(dead, a.err, lane);
  DF_TIMING_REPORT("backward")
#undef P_DONE
#undef T_DONE
#undef N_OF
#undef MAC_NP
}
#undef DFC
#define DFC(j) (j)

// ---- maximum-accuracy DP ----------------------------------------------------------------------------------------------
template <bool LOCAL, bool GROWS, int DP_AHEAD>
__global__ void __launch_bounds__(64) hhv_mac_dp_kernel(MacArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* S = GROWS ? reinterpret_cast<float*>(a.row_scratch + (size_t)blockIdx.x * 10 * (a.lds_cols + 2))
                   : reinterpret_cast<float*>(smem);  // [2][Lt+2]
  const int k = a.sel[blockIdx.x], lane = threadIdx.x;
  const HitView h = view(a, k);
  const int Lq = h.Lq, Lt = h.Lt, pitch = h.pitch, stride = Lt + 2;
  const float mact = a.mact;
  const double half = 0.5 * mact;
  for (int e = lane; e < 2 * stride; e += 64) S[e] = 0.0f;
  for (int e = lane; e < pitch; e += 64) h.bmm[e] = MAC_STOP;
  __syncthreads();
  int cur = 0;
  float best = -FLT_MAX;
  int best_i = 0, best_j = 0;
  // Mask byte and posterior of a strip are fetched DP_AHEAD strips before they are used (also across rows).  One strip ahead
  // (rounds 2-4) is right where every strip of a row has active cells (500 hits of 300 columns: 0.83 ms; four ahead 0.98 ms -
  // the extra register traffic of a lone, latency-bound wave); in long templates most strips of a row are masked out and take
  // the wave a few hundred clocks, less than a trip to L2: four ahead took a batch of mixed lengths (40 .. 1 800 columns) from
  // 14.4 to 11.9 ms.  The launcher picks by length class.  pf_i / pf_s0: the strip the next fetch is for.
  unsigned char co_q[DP_AHEAD];
  float p_q[DP_AHEAD];
  int pf_i = 1, pf_s0 = 0;
  auto fetch_next = [&](unsigned char& co, float& p) {
    const int ni = min(pf_i, Lq), nj = min(1 + pf_s0 + lane, Lt);
    co = h.co[(size_t)ni * pitch + nj];
    p = h.mat[(size_t)ni * pitch + nj];
    pf_s0 += 64;
    if (pf_s0 >= Lt) pf_s0 = 0, ++pf_i;
  };
#pragma unroll
  for (int e = 0; e < DP_AHEAD; ++e) fetch_next(co_q[e], p_q[e]);
  for (int i = 1; i <= Lq; ++i) {
    const int prv = cur ^ 1;
    const float* Sp = S + prv * stride;
    float* Sc = S + cur * stride;
    float carry = 0.0f;  // S_curr[0]
    for (int s0 = 0; s0 < Lt; s0 += 64) {
      const int j = 1 + s0 + lane;
      const bool valid = j <= Lt;
      const int jc = valid ? j : Lt;
      const bool off = co_q[0] != 0;
      const float p = p_q[0];
#pragma unroll
      for (int e = 0; e + 1 < DP_AHEAD; ++e) co_q[e] = co_q[e + 1], p_q[e] = p_q[e + 1];
      fetch_next(co_q[DP_AHEAD - 1], p_q[DP_AHEAD - 1]);
      const float term1 = p - mact;
      const float term2 = Sp[jc - 1] + p - mact;
      const float term3 = (float)(Sp[jc] - half);
      float mx;
      int code;
      if (term1 > term2) {
        mx = term1;
        code = MAC_STOP;
      } else {
        mx = term2;
        code = MAC_MM;
      }
      if (term3 > mx) {
        mx = term3;
        code = MAC_MI;
      }
      // inactive cells hold -FLT_MIN whatever their neighbour says: sweep only from the first to the last active lane.
      // The chain step is S = max(mx, (float)((double)S_left - 0.5*mact)).  0.5*mact is a float (halving is exact) and
      // rounding a difference of two floats first to double and then to float equals rounding it once (53 >= 2*24+2
      // bits), so the step is one v_sub_f32; the maximum and the cell-off override are one v_med3_f32:
      // med3(t4, mx, +inf) = max(t4, mx), med3(t4, -FLT_MIN, -FLT_MIN) = -FLT_MIN.
      const unsigned long long on_mask = __ballot(!off && valid);
      const int n_steps = on_mask ? (63 - __builtin_clzll(on_mask)) - __builtin_ctzll(on_mask) + 1 : 0;
      const float half_f = (float)half;
      const float lo = off ? -FLT_MIN : mx, hi = off ? -FLT_MIN : __builtin_inff();
      float sv = off ? -FLT_MIN : 0.0f;
      if (__ballot(mx != mx) == 0) {
        // lane 0's left neighbour is the carry, a constant of the sweep: its value is fixed up front (med3(x, v, v) = v) and
        // the loop shifts zeros into lane 0 - the step is then v_sub_f32 with a DPP operand + v_med3_f32
        const float v0 = __builtin_amdgcn_fmed3f(carry - half_f, lo, hi);
        const float lo_s = lane == 0 ? v0 : lo, hi_s = lane == 0 ? v0 : hi;
        for (int s = 0; s < n_steps; ++s) sv = __builtin_amdgcn_fmed3f(shr1_fz(sv) - half_f, lo_s, hi_s);
      } else {  // NaN posteriors (a mask that leaves no path: Pforward = 0): the literal compare/select chain
        for (int s = 0; s < n_steps; ++s) {
          const float t4s = shr1_f(sv, carry) - half_f;
          sv = off ? -FLT_MIN : (t4s > mx ? t4s : mx);
        }
      }
      const float t4 = shr1_f(sv, carry) - half_f;
      if (off)
        code = MAC_STOP;
      else if (t4 > mx)
        code = MAC_IM;
      if (valid) {
        Sc[j] = sv;
        h.bmm[(size_t)i * pitch + j] = (unsigned char)code;
        if (!off && sv > best && (LOCAL || i == Lq)) {
          best = sv;
          best_i = i;
          best_j = j;
        }
      }
      carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sv), 63));
    }
    if (lane == 0) h.bmm[(size_t)i * pitch] = 0;
    __syncthreads();
    if (!LOCAL && lane == 0) {
      // global alignment: best cell of the last column (:146-151); lane 0 owns column Lt's bookkeeping
      const float v = Sc[Lt];
      if (v > best) {
        best = v;
        best_i = i;
        best_j = Lt;
      }
    }
    cur = prv;
  }
  // first maximum in the reference's visiting order: largest value, then smallest (i, j) -- for the global mode the
  // visiting order is (1,Lt) .. (Lq-1,Lt), (Lq,1..Lt): rows < Lq only ever hold column Lt, so (i, j) order is the same
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64), oj = __shfl_xor(best_j, o, 64);
    const bool take = ov > best || (ov == best && (oi < best_i || (oi == best_i && oj < best_j)));
    if (take) {
      best = ov;
      best_i = oi;
      best_j = oj;
    }
  }
  if (lane == 0) {
    a.hits[k].i2 = best_i;
    a.hits[k].j2 = best_j;
  }
}

// ---- per-row active ranges ---------------------------------------------------------------------------------------------------
// rng[i] = (first, last) template column of query row i whose mask byte is 0 (first > last: none), rows 1 .. Lq of every hit.
// hhv_mac_dp_diag_kernel visits only the hull of the ranges of its 64 rows, and the MAC backtrace takes cells outside their row's
// range as STOP (what the reference leaves in every masked cell, src/hhmacalgorithm.cpp:66-70) without reading the code plane.
// Any SUPERSET of the active cells will do (the kernels still read the mask bytes inside it): masks built on the device get
// theirs from the geometry (hhv_mac_mask_kernel), this kernel scans masks handed over by the host.
__global__ void __launch_bounds__(256) hhv_mac_rowrange_kernel(MacArgs a) {
  const int k = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Lq = a.Lq, Lt = a.Lt[k], pitch = Lt + 1;
  const unsigned char* co = a.celloff + a.mat_off[k];
  int2* rng = a.row_rng + (size_t)k * (Lq + 2);
  if (Lt >= a.sparse_min_Lt) {  // sparse rows: the cells nobody visits must read 0 (StripSpan)
    float* mat = a.mat + a.mat_off[k];
    for (int c = threadIdx.x; c < (Lq + 1) * pitch; c += 256) mat[c] = 0.0f;
  }
  for (int i = wave; i <= Lq; i += 4) {
    int lo = 0x7fffffff, hi = 0;
    for (int j0 = 1; j0 <= Lt && i >= 1; j0 += 64) {
      const int j = j0 + lane;
      const unsigned long long m = __ballot(j <= Lt && co[(size_t)i * pitch + j] == 0);
      if (m) {
        lo = min(lo, j0 + (int)__builtin_ctzll(m));
        hi = max(hi, j0 + 63 - (int)__builtin_clzll(m));
      }
    }
    if (lane == 0) rng[i] = make_int2(lo, hi);
  }
  __syncthreads();
  {
    __shared__ int s_span;  // the widest row in strips (see hhv_mac_mask_kernel)
    if (threadIdx.x == 0) s_span = 0;
    __syncthreads();
    int mx = 0;
    for (int i = 1 + (int)threadIdx.x; i <= Lq; i += 256)
      mx = max(mx, mac_row_span(i > 1 ? rng[i - 1] : make_int2(1, 0), rng[i], i < Lq ? rng[i + 1] : make_int2(1, 0), Lt));
    atomicMax(&s_span, mx);
    __syncthreads();
    if (threadIdx.x == 0) rng[0] = make_int2(s_span, 0);
  }
}

// ---- maximum-accuracy DP along anti-diagonals ----------------------------------------------------------------------------------
// src/hhmacalgorithm.cpp:53-156.  Unlike forward / backward this recurrence has no per-row rescaling, so nothing forces rows
// to be finished one after the other: the 64 lanes of the wavefront take 64 CONSECUTIVE QUERY ROWS (a strip) and walk along the
// template, lane l one column behind lane l-1.  In step t lane l evaluates cell (i0 + 1 + l, jlo + t - l) from
//   up   = S(i-1, j)    the value lane l-1 produced in the step before (one DPP move; lane 0: the last row of the strip above, Sb)
//   diag = S(i-1, j-1)  the `up` of the step before
//   left = S(i, j-1)    its own value of the step before
// with the reference's float expressions and comparisons, literally (NaN posteriors included).  A strip costs Wd + 63 steps of
// some fifty instructions instead of Lt/64 sweeps of up to 64 dependent steps per row (hhv_mac_dp_kernel above).
// [jlo, jhi] = hull of the active ranges of the strip's rows (hhv_mac_rowrange_kernel): cells outside are masked in all 64
// rows (S = -FLT_MIN, code STOP - already in the plane) and are not visited at all - a 300 x 1800 hit whose band is 200 wide
// costs 5 x 263 steps.
// Memory: a lane's cell is in another row AND column than its neighbour's, so posteriors / mask bytes / codes go through LDS
// tiles [64 rows][2 x 64 columns]: every fourth step FOUR rows of a column block are fetched by one load (16 lanes a row, 16
// bytes of posteriors / 4 mask bytes a lane, 64 steps before their lanes get there), parked in LDS eight steps later; every lane
// reads its own row at its own column (row pitch - 1 odd: no bank conflict); codes take the way back, four rows per store.
// Sb: S of the last row of the strip above, per column.
constexpr int DPD_PITCH = 132;   // floats per row of the posterior tile (rows 16-byte aligned)
constexpr int DPD_BPITCH = 132;  // bytes per row of the mask and code tiles
constexpr int DPD_AHEAD = 8;     // steps between a row's global load and its store into the tile (the loop is unrolled by it)
constexpr size_t DPD_TILES = (size_t)64 * DPD_PITCH * sizeof(float) + (size_t)2 * 64 * DPD_BPITCH;
template <bool LOCAL, bool GROWS>
__global__ void __launch_bounds__(64) hhv_mac_dp_diag_kernel(MacArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* PT = reinterpret_cast<float*>(smem);
  unsigned char* CO = smem + (size_t)64 * DPD_PITCH * sizeof(float);
  unsigned char* CD = CO + (size_t)64 * DPD_BPITCH;
  float* Sb = GROWS ? reinterpret_cast<float*>(a.row_scratch + (size_t)blockIdx.x * 10 * (a.lds_cols + 2))
                    : reinterpret_cast<float*>(smem + DPD_TILES);  // [Lt + 2]
  const int k = a.sel[blockIdx.x], lane = threadIdx.x;
  const HitView h = view(a, k);
  const int Lq = h.Lq, Lt = h.Lt, pitch = h.pitch;
  const int2* rng = a.row_rng + (size_t)k * (Lq + 2);
  const float mact = a.mact;
  const float half_f = (float)(0.5 * (double)mact);  // exact (the launcher sends other values of mact to hhv_mac_dp_kernel)
  for (int e = lane; e <= Lt + 1; e += 64) Sb[e] = 0.0f;  // row 0 (:44-45)
  float best = -FLT_MAX;
  int best_i = 0, best_j = 0;
  for (int i0 = 0; i0 < Lq; i0 += 64) {
    const int nr = min(64, Lq - i0);
    const int my_i = i0 + 1 + lane;
    const bool has_row = lane < nr;
    int lo = 0x7fffffff, hi = 0;
    if (has_row) {
      const int2 r = rng[my_i];
      lo = r.x;
      hi = r.y;
    }
    for (int o = 32; o >= 1; o >>= 1) {
      lo = min(lo, __shfl_xor(lo, o, 64));
      hi = max(hi, __shfl_xor(hi, o, 64));
    }
    const int jlo = __builtin_amdgcn_readfirstlane(lo), jhi = __builtin_amdgcn_readfirstlane(hi);
    const bool none = jlo > jhi;  // no active cell in the whole strip
    const int Wd = none ? 0 : jhi - jlo + 1;
    __syncthreads();  // (GROWS: Sb is global memory, written and read by this one wavefront)
    // S(i0, jlo - 1) for lane 0, before this strip's last row takes the place of the one above
    const float diag0 = none ? 0.0f : Sb[jlo - 1];
    __syncthreads();
    for (int j = 1 + lane; j <= Lt; j += 64)
      if (none || j < jlo || j > jhi) Sb[j] = -FLT_MIN;  // the last row of THIS strip outside the hull: masked cells (:66-67)
    if (!none) {
      const float v0 = jlo == 1 ? 0.0f : -FLT_MIN;  // S(i, jlo - 1): column 0 (:58) or a masked cell
      float S = v0, diag = lane == 0 ? diag0 : v0, sbv = 0.0f;
      const int Wd_l = has_row ? Wd : 0;
      const int cl = (Wd - 1) >> 6;
      const int t_end = 64 * cl + 136;  // exclusive (the last codes leave at step 64 cl + 128); t_end + 64: a multiple of DPD_AHEAD
      const size_t row0 = (size_t)(i0 + 1) * pitch + jlo;
      const int my_tile = lane * DPD_PITCH, my_btile = lane * DPD_BPITCH;
      const int g_row = lane >> 4, g_q4 = (lane & 15) * 4;  // a group of four rows: 16 lanes a row, four columns a lane
      float4 qp[DPD_AHEAD / 4];
      uint32_t qc[DPD_AHEAD / 4];
#pragma unroll
      for (int e = 0; e < DPD_AHEAD / 4; ++e) qp[e] = make_float4(0.0f, 0.0f, 0.0f, 0.0f), qc[e] = 0x01010101u;
      for (int tb = -64; tb < t_end; tb += DPD_AHEAD) {
#pragma unroll
        for (int e = 0; e < DPD_AHEAD; ++e) {
          const int t = tb + e;
          const bool in_dp = t >= 0 && t < Wd + 63;
          const int x = t - lane, xi = x & 127;
          float p = 0.0f;
          bool off = true;
          if (in_dp) {  // this step's posterior and mask byte (parked at least 56 steps ago): issued first, used last
            p = PT[my_tile + xi];
            off = CO[my_btile + xi] != 0;
            if ((t & 63) == 0) sbv = Sb[min(jlo + t + lane, Lt)];  // the row above, 64 columns at a time
          }
          if ((e & 3) == 0) {
            constexpr int G = 0;
            const int g = (e >> 2) + G;
            {  // park the four rows fetched DPD_AHEAD steps ago
              const int u = t + 64 - DPD_AHEAD;
              if (u >= 0) {
                const int r = (u & 63) + g_row, c = u >> 6, slot = (c & 1) * 64;
                *reinterpret_cast<float4*>(&PT[r * DPD_PITCH + slot + g_q4]) = qp[g];
                const int nv = Wd - (c * 64 + g_q4);  // columns of this lane inside the hull; the others read as masked
                const uint32_t beyond = nv >= 4 ? 0u : (nv <= 0 ? 0x01010101u : 0x01010101u << (8 * nv));
                *reinterpret_cast<uint32_t*>(&CO[r * DPD_BPITCH + slot + g_q4]) = qc[g] | beyond;
              }
            }
            {  // fetch rows r .. r+3 of column block c: 64 steps before lane r starts on it
              const int u = t + 64;
              const int r = (u & 63) + g_row, c = u >> 6;
              if (c <= cl && r < nr) {
                const size_t idx = row0 + (size_t)r * pitch + c * 64 + g_q4;
                __builtin_memcpy(&qp[g], h.mat + idx, 16);  // (rows start at any multiple of four bytes: dwordx4 needs no more)
                __builtin_memcpy(&qc[g], h.co + idx, 4);    // (a dword at any address: the target's unaligned access mode)
              }
            }
            if (t >= 68) {  // codes of rows rf .. rf+3, column block cf: complete since two steps ago
              const int tf = t - 4;
              const int rf = (tf & 63) + g_row, cf = (tf >> 6) - 1, xg = cf * 64 + g_q4;
              if (rf < nr && xg < Wd) {
                const uint32_t cv = *reinterpret_cast<const uint32_t*>(&CD[rf * DPD_BPITCH + (cf & 1) * 64 + g_q4]);
                unsigned char* dst = h.bmm + row0 + (size_t)rf * pitch + xg;
                if (xg + 3 < Wd) {
                  __builtin_memcpy(dst, &cv, 4);
                } else {  // the hull's last columns: byte by byte (the bytes behind them belong to the next row)
                  for (int q = 0; q < Wd - xg; ++q) dst[q] = (unsigned char)(cv >> (8 * q));
                }
              }
            }
          }
          if (in_dp) {
            const bool active = (unsigned)x < (unsigned)Wd_l;
            const float sb_t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sbv), t & 63));
            const float up = shr1_f(S, sb_t);
            const float term1 = p - mact;            // :76
            const float term2 = (diag + p) - mact;   // :78
            const float term3 = up - half_f;         // :79 (a float difference rounded once, see hhv_mac_dp_kernel)
            const float term4 = S - half_f;          // :80
            float mx;
            int code;
            if (term1 > term2) {
              mx = term1;
              code = MAC_STOP;
            } else {
              mx = term2;
              code = MAC_MM;
            }
            if (term3 > mx) {
              mx = term3;
              code = MAC_MI;
            }
            if (term4 > mx) {
              mx = term4;
              code = MAC_IM;
            }
            if (off) {
              mx = -FLT_MIN;
              code = MAC_STOP;
            }
            diag = up;
            if (active) {
              S = mx;
              CD[my_btile + xi] = (unsigned char)code;
              const int j = jlo + x;
              // :123-126 inside the row, :146-151 the last column of a global alignment (whatever its mask byte says)
              const bool cand = LOCAL ? !off : ((my_i == Lq && !off) || j == Lt);
              if (cand && mx > best) {
                best = mx;
                best_i = my_i;
                best_j = j;
              }
              if (lane == 63) Sb[j] = mx;
            }
          }
        }
      }
    }
    if (!LOCAL && has_row && (none || jhi < Lt)) {
      // global alignment: S(i, Lt) of a row whose last column lies outside the hull is -FLT_MIN (:146-151)
      if (-FLT_MIN > best) {
        best = -FLT_MIN;
        best_i = my_i;
        best_j = Lt;
      }
    }
  }
  // first maximum in the reference's visiting order: largest value, then smallest (i, j) (see hhv_mac_dp_kernel)
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(best_i, o, 64), oj = __shfl_xor(best_j, o, 64);
    const bool take = ov > best || (ov == best && (oi < best_i || (oi == best_i && oj < best_j)));
    if (take) {
      best = ov;
      best_i = oi;
      best_j = oj;
    }
  }
  if (lane == 0) {
    a.hits[k].i2 = best_i;
    a.hits[k].j2 = best_j;
  }
}

// ---- MAC backtrace: one wavefront per hit --------------------------------------------------------------------------------
// The walk itself is a pointer chase (src/hhbacktracemac.cpp:126-160) and every step used to cost one dependent trip to
// L2 / HBM for a single code byte.  Now the wave fetches the 8 x 8 block of codes above-left of the current cell with one
// load (lane l holds cell (i - l/8, j - l%8)) and walks inside it with v_readlane until it leaves the block: eight or more
// steps per memory latency.  The per-step column scores and posteriors are then computed by all lanes, 64 steps at a time;
// only the sum of the posteriors keeps the reference's sequential order (:196-197).
__global__ void __launch_bounds__(64) hhv_mac_trace_kernel(MacArgs a) {
  const int k = blockIdx.x, lane = threadIdx.x;
  const HitView h = view(a, k);
  const int pitch = h.pitch;
  int* is = a.path_i + a.path_off[k];
  int* js = a.path_j + a.path_off[k];
  signed char* st = a.path_state + a.path_off[k];
  float* Ss = a.path_S + a.path_off[k];
  float* Ps = a.path_P + a.path_off[k];
  int i = __builtin_amdgcn_readfirstlane(a.hits[k].i2), j = __builtin_amdgcn_readfirstlane(a.hits[k].j2);
  // the block of codes whose bottom-right corner is (bi, bj); :124-125: b[i][1] = b[1][j] = STOP
  int bi = i, bj = j, codes;
  const int2* rng = a.row_rng + (size_t)k * (h.Lq + 2);
  auto load_block = [&]() {
    const int ti = bi - (lane >> 3), tj = bj - (lane & 7);
    // (cells outside the active range of their row are masked: STOP, and the DP kernel has not written their code)
    const bool border = ti < 1 || tj < 1 || ti == 1 || tj == 1;
    const int2 r = border ? make_int2(1, 0) : rng[ti];
    codes = (border || tj < r.x || tj > r.y) ? (int)MAC_STOP : (int)h.bmm[(size_t)ti * pitch + tj];
  };
  load_block();
  int step = 0, state = MAC_MM, matched = 1;
  if (__builtin_amdgcn_readlane(codes, 0) != MAC_MM) {
    if (lane == 0) {
      is[0] = i;
      js[0] = j;
    }
  } else {
    while (state != MAC_STOP) {
      if (bi - i > 7 || bj - j > 7) {
        bi = i;
        bj = j;
        load_block();
      }
      step++;
      state = __builtin_amdgcn_readlane(codes, (bi - i) * 8 + (bj - j));
      if (lane == 0) {
        st[step] = (signed char)state;
        is[step] = i;
        js[step] = j;
      }
      if (state == MAC_MM) matched++;
      switch (state) {
        case MAC_MM: i--; j--; break;
        case MAC_IM: j--; break;
        case MAC_MI: i--; break;
        case MAC_STOP: break;
        default: state = 0; break;
      }
    }
  }
  if (lane == 0) {
    st[step] = MAC_MM;  // :170
    if (step > 0) {       // entry 0 is unused then (the reference leaves it uninitialised); keep it deterministic
      is[0] = js[0] = 0;
      st[0] = 0;
    }
    Ss[0] = Ps[0] = 0.0f;
  }
  __threadfence();  // the path written by lane 0 is read by all lanes below
  // per-step scores (:183-203), 64 steps at a time
  float sum = 0.0f;
  const int i1 = step > 0 ? is[step] : is[0], j1 = step > 0 ? js[step] : js[0];
  for (int s0 = 1; s0 <= step; s0 += 64) {
    const int sidx = s0 + lane;
    const bool in = sidx <= step;
    const bool mm = in && st[sidx] == MAC_MM;
    float S = 0.0f, P = 0.0f;
    if (mm) {
      const int si = is[sidx], sj = js[sidx];
      S = fast_log2_mac(dot20(h.qp + (size_t)si * 20, h.tp + (size_t)sj * h.tps), a.lg2, a.diff);
      P = h.mat[(size_t)si * pitch + sj];
    }
    if (in) {
      Ss[sidx] = S;
      Ps[sidx] = P;
    }
    const unsigned long long mm_mask = __ballot(mm);
    const int cnt = min(64, step - s0 + 1);
    for (int l = 0; l < cnt; ++l)  // the reference's sequential float sum over the match states, in step order
      if ((mm_mask >> l) & 1ull) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(P), l));
  }
  if (lane == 0) {
    a.hits[k].nsteps = step;
    a.hits[k].matched_cols = matched;
    a.hits[k].i1 = i1;
    a.hits[k].j1 = j1;
    a.hits[k].sum_of_probs = sum;
    a.hits[k].Pforward = a.Pforward[k];
    a.hits[k].pad = 0;
  }
}

#undef ROW

// ---- the cell-off mask realign() builds (src/hhposteriordecoder.cpp:92-109), one workgroup per hit ------------------
//   maskViterbiAlignment (:205-240): everything off except the rectangles above-left of (i1,j1) and below-right of
//   (i2,j2); then +-40 rows / columns around every step of the Viterbi path on;   excludeMACAlignment (:245-262): the
//   +-2 cross of every cell of the earlier MAC alignments off;   exclude_regions / exclude_template_regions (:121-149).
// (Viterbi::InitializeForAlignment's minimum-overlap corners are overwritten by maskViterbiAlignment and left out.)
// The kernel also leaves rng[i] (hhv_mac_rowrange_kernel) for every row, from the geometry: the two rectangles and the band are a
// superset of the cells it switches on (the excluded cells and ranges only switch cells off).  LDS_RNG: the per-row minima and
// maxima of the band are collected with LDS atomics (queries up to MAC_MASK_LDS_ROWS rows), else with global ones.
constexpr int MAC_MASK_LDS_ROWS = 8000;
template <bool LDS_RNG>
__global__ void __launch_bounds__(256) hhv_mac_mask_kernel(MacArgs a, MacMaskArgs m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* s_lo = reinterpret_cast<int*>(smem);  // [Lq + 1]
  int* s_hi = s_lo + (a.Lq + 1);
  const int k = blockIdx.x, tid = threadIdx.x;
  const int Lq = a.Lq, Lt = a.Lt[k], pitch = Lt + 1;
  unsigned char* co = const_cast<unsigned char*>(a.celloff) + a.mat_off[k];
  int2* rng = a.row_rng + (size_t)k * (Lq + 2);
  // the Viterbi alignment of the hit: handed over by the host, or - resident hits - taken from the trace results of the
  // template set the Viterbi stage searched (hhv_hits)
  int4 e = m.ends[k];  // i1, j1, i2, j2
  const int32_t* vi = m.vit_i + m.vit_off[k];
  const int32_t* vj = m.vit_j + m.vit_off[k];
  int ns = (int)(m.vit_off[k + 1] - m.vit_off[k]);
  if (m.res_hits && m.res_template[k] >= 0) {
    const int t = m.res_template[k];
    const DevHit h = m.res_hits[t];
    e = make_int4(h.i1, h.j1, h.i2, h.j2);
    ns = h.nsteps;
    vi = m.res_i + m.res_path_off[t] + 1;  // entries 1..nsteps
    vj = m.res_j + m.res_path_off[t] + 1;
  }
  // rows' ranges, first from the two rectangles
  for (int i = tid; i <= Lq; i += 256) {
    int lo = 0x7fffffff, hi = 0;
    if (i >= 1 && i < e.x && e.y > 1) lo = 1, hi = min(e.y - 1, Lt);
    if (i >= 1 && i > e.z && e.w < Lt) lo = min(lo, max(e.w + 1, 1)), hi = Lt;
    if (LDS_RNG) s_lo[i] = lo, s_hi[i] = hi;
    else rng[i] = make_int2(lo, hi);
  }
  // the plane four cells (one dword: the planes start at multiples of four bytes and are padded to one) at a time
  const int cells = (Lq + 1) * pitch;
  uint32_t* co4 = reinterpret_cast<uint32_t*>(co);
  for (int c = 4 * tid; c < cells; c += 4 * 256) {
    int i = c / pitch, j = c - i * pitch;
    uint32_t v = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t off = (i >= 1 && j >= 1 && i <= Lq) ? !((i < e.x && j < e.y) || (i > e.z && j > e.w)) : 0;
      v |= off << (8 * q);
      if (++j == pitch) j = 0, ++i;
    }
    co4[c >> 2] = v;
  }
  if (Lt >= a.sparse_min_Lt) {  // sparse rows: the cells nobody visits must read 0 (StripSpan)
    float4* mat4 = reinterpret_cast<float4*>(a.mat + a.mat_off[k]);  // (the planes start at multiples of four elements and are padded to one)
    for (int c = tid; c < (cells + 3) / 4; c += 256) mat4[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  __syncthreads();
  constexpr int W = 2 * 40 + 1;  // FWD_BKW_PATHWITDH = 40 (src/hhdecl.h:37)
  for (int w = tid; w < ns * W; w += 256) {
    const int step = w / W, d = w - step * W - 40;
    const int pi = vi[step], pj = vj[step];
    if (pi + d >= 1 && pi + d <= Lq && pj >= 1 && pj <= Lt) {
      co[(size_t)(pi + d) * pitch + pj] = 0;
      if (LDS_RNG) {
        atomicMin(&s_lo[pi + d], pj);
        atomicMax(&s_hi[pi + d], pj);
      } else {
        atomicMin(&rng[pi + d].x, pj);
        atomicMax(&rng[pi + d].y, pj);
      }
    }
    if (pj + d >= 1 && pj + d <= Lt && pi >= 1 && pi <= Lq) {
      co[(size_t)pi * pitch + pj + d] = 0;
      if (d == -40 || d == 40 || pj + d == 1 || pj + d == Lt) {  // the ends of the row segment
        if (LDS_RNG) {
          atomicMin(&s_lo[pi], pj + d);
          atomicMax(&s_hi[pi], pj + d);
        } else {
          atomicMin(&rng[pi].x, pj + d);
          atomicMax(&rng[pi].y, pj + d);
        }
      }
    }
  }
  __syncthreads();
  if (LDS_RNG)
    for (int i = tid; i <= Lq; i += 256) rng[i] = make_int2(s_lo[i], s_hi[i]);
  __syncthreads();
  {
    // the widest row of the hit, in strips (mac_row_span): rng[0].x - which kind of kernel runs the hit when its template is too
    // long for the plain LDS layout (RING)
    __shared__ int s_span;
    if (tid == 0) s_span = 0;
    __syncthreads();
    int mx = 0;
    for (int i = 1 + tid; i <= Lq; i += 256)
      mx = max(mx, mac_row_span(i > 1 ? rng[i - 1] : make_int2(1, 0), rng[i], i < Lq ? rng[i + 1] : make_int2(1, 0), Lt));
    atomicMax(&s_span, mx);
    __syncthreads();
    if (tid == 0) rng[0] = make_int2(s_span, 0);
  }
  const int64_t x0 = m.excl_off[k];
  const int nx = (int)(m.excl_off[k + 1] - x0);
  for (int w = tid; w < nx * 5; w += 256) {
    const int c = w / 5, d = w - c * 5 - 2;
    const int pi = m.excl_i[x0 + c], pj = m.excl_j[x0 + c];
    if (pi + d >= 1 && pi + d <= Lq && pj >= 0 && pj <= Lt) co[(size_t)(pi + d) * pitch + pj] = 1;
    if (pj + d >= 1 && pj + d <= Lt && pi >= 0 && pi <= Lq) co[(size_t)pi * pitch + pj + d] = 1;
  }
  for (int r = 0; r < m.n_qranges; ++r) {
    const int lo = m.ranges[2 * r], hi = min(m.ranges[2 * r + 1], Lq);
    for (int c = tid; c < (hi - lo + 1) * Lt; c += 256) {
      const int i = lo + c / Lt, j = 1 + c % Lt;
      if (i >= 0) co[(size_t)i * pitch + j] = 1;
    }
  }
  for (int r = 0; r < m.n_tranges; ++r) {
    const int lo = m.ranges[2 * (m.n_qranges + r)], hi = min(m.ranges[2 * (m.n_qranges + r) + 1], Lt);
    for (int c = tid; c < (hi - lo + 1) * Lq; c += 256) {
      const int j = lo + c / Lq, i = 1 + c % Lq;
      if (j >= 0) co[(size_t)i * pitch + j] = 1;
    }
  }
  __syncthreads();
  for (int c = tid; c <= Lt; c += 256) co[c] = 0;  // row / column 0 are never read
  for (int c = tid; c <= Lq; c += 256) co[(size_t)c * pitch] = 0;
}

int launch_mac_mask(const MacArgs& a, const MacMaskArgs& m, void* stream) {
  if (a.Lq <= MAC_MASK_LDS_ROWS)
    hipLaunchKernelGGL(hhv_mac_mask_kernel<true>, dim3(a.n), dim3(256), (size_t)2 * (a.Lq + 1) * sizeof(int), (hipStream_t)stream, a, m);
  else
    hipLaunchKernelGGL(hhv_mac_mask_kernel<false>, dim3(a.n), dim3(256), 0, (hipStream_t)stream, a, m);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
// masks handed over by the host: their rows' ranges by a scan
int launch_mac_rowrange(const MacArgs& a, void* stream) {
  hipLaunchKernelGGL(hhv_mac_rowrange_kernel, dim3(a.n), dim3(256), 0, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

// LDS of the forward / backward kernels: two rows of state, plus - when it fits - the template itself
size_t mac_rows_lds(int max_Lt, bool stage) {
  // the dataflow kernels: 14 doubles per column + the control block; staged: + the template (28 floats per column) + two rows of
  // mask bytes + two rows of F_MM fetched a row ahead + the secondary-structure table of the hit
  return ((size_t)MAC_ROW_FIELDS * (max_Lt + 2) + MAC_CTL_DOUBLES) * sizeof(double) +
         (stage ? (size_t)(max_Lt + 2) * 28 * sizeof(float) + (((size_t)2 * (max_Lt + 2) + 15) & ~(size_t)15) +
                      (size_t)2 * (max_Lt + 2) * sizeof(float) + 352 * sizeof(float) : 0);
}
constexpr size_t MAC_LDS_LIMIT = 160 * 1024;

template <bool LOCAL, bool STAGE, bool GROWS>
static void launch_mac_rows(const MacArgs& a, int n, size_t lds, hipStream_t stream, bool dataflow) {
  if constexpr (!GROWS && !STAGE) {
    if (dataflow) {
      // row state in LDS: the dataflow kernels (eight wavefronts per hit), template operands from global memory
      (void)hipFuncSetAttribute((const void*)hhv_mac_forward_df_kernel<LOCAL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      static const bool dbg = getenv("HHV_MAC_DEBUG") != nullptr;  // measurement aid: resident workgroups per CU
      if (dbg) {
        int nf = 0, nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nf, (const void*)hhv_mac_forward_df_kernel<LOCAL, false>, MAC_DF_THREADS, lds);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)hhv_mac_backward_df_kernel<LOCAL, false, false>, MAC_DFB_THREADS, lds);
        fprintf(stderr, "hhv_mac: %d hits, %zu B of LDS per workgroup: %d forward / %d backward workgroups resident per CU\n", n, lds, nf, nb);
      }
      hipLaunchKernelGGL((hhv_mac_forward_df_kernel<LOCAL, false>), dim3(n), dim3(MAC_DF_THREADS), lds, stream, a);
      if (a.fwd_list) {  // the -o_matrices lists were asked for (hhv_mac_set_lists)
        hipLaunchKernelGGL(hhv_mac_fwdlist_kernel, dim3(n), dim3(256), 0, stream, a);
        (void)hipFuncSetAttribute((const void*)hhv_mac_backward_df_kernel<LOCAL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((hhv_mac_backward_df_kernel<LOCAL, false, true>), dim3(n), dim3(MAC_DFB_THREADS), lds, stream, a);
      } else {
        (void)hipFuncSetAttribute((const void*)hhv_mac_backward_df_kernel<LOCAL, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((hhv_mac_backward_df_kernel<LOCAL, false, false>), dim3(n), dim3(MAC_DFB_THREADS), lds, stream, a);
      }
      return;
    }
  }
  (void)hipFuncSetAttribute((const void*)hhv_mac_forward_kernel<LOCAL, STAGE, GROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((hhv_mac_forward_kernel<LOCAL, STAGE, GROWS>), dim3(n), dim3(64), lds, stream, a);
  if (a.fwd_list) {
    hipLaunchKernelGGL(hhv_mac_fwdlist_kernel, dim3(n), dim3(256), 0, stream, a);
    (void)hipFuncSetAttribute((const void*)hhv_mac_backward_kernel<LOCAL, STAGE, GROWS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((hhv_mac_backward_kernel<LOCAL, STAGE, GROWS, true>), dim3(n), dim3(64), lds, stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)hhv_mac_backward_kernel<LOCAL, STAGE, GROWS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((hhv_mac_backward_kernel<LOCAL, STAGE, GROWS, false>), dim3(n), dim3(64), lds, stream, a);
  }
}
// aux / aux_join: a second stream (already waiting for the inputs) and its join event for the class whose hits are of two kinds
template <bool LOCAL>
static void launch_mac_class(const MacArgs& a, int cls, int n, int max_Lt, hipStream_t stream, hipStream_t aux = nullptr,
                             hipEvent_t aux_join = nullptr, int n_long = 0) {
  static_assert(MAC_PRE <= MAC_DF_STRIPS, "staged classes: at most MAC_PRE strips a row");
  // classes 0 .. 5: the dataflow kernels with the template operands read from global memory (round 6: the copy of the template in LDS
  // the single-wave kernels needed - STAGE - costs the dataflow kernels twice the footprint and is no faster: 500 hits 300 x 300
  // 3.48 -> 3.32 ms without it, mixed lengths 6.2 -> 5.2 ms, where the LDS of the CUs is what limits the hits in flight).
  // HHV_MAC_NO_PIPE (measurement aid): the single-wave kernels for every class, staged where the template fits.
  static const bool no_pipe = getenv("HHV_MAC_NO_PIPE") != nullptr;
  if (cls <= 5 && !no_pipe) launch_mac_rows<LOCAL, false, false>(a, n, mac_rows_lds(max_Lt, false), stream, true);
  else if (cls <= 3 && mac_rows_lds(max_Lt, true) <= MAC_LDS_LIMIT && max_Lt <= MAC_PRE * 64) launch_mac_rows<LOCAL, true, false>(a, n, mac_rows_lds(max_Lt, true), stream, false);
  else if (cls <= 5) launch_mac_rows<LOCAL, false, false>(a, n, mac_rows_lds(max_Lt, false), stream, false);  // (fits: mac_length_class)
  else {
    // templates beyond the plain LDS layout: the dataflow kernels on a ring of strips for the hits whose rows all fit it, the
    // single-wave kernels (row state in global memory) for the others - both launched over the whole class, a workgroup whose
    // hit is of the other kind returns at once (the kind is decided on the device: rng[0].x, written with the masks)
    // (the ring kernels over the first n_long hits of the class only - the class lists its longest templates first -, the others
    // are short templates beyond the dataflow budget: 1 500 workgroups asking for a whole CU's LDS just to return would wait for it)
    static const bool ring_off = getenv("HHV_MAC_NO_RING") != nullptr || getenv("HHV_MAC_NO_PIPE") != nullptr;  // measurement aids
    const bool no_ring = ring_off || n_long <= 0;
    const int n_ring = n_long < n ? n_long : n;
    MacArgs ar = a, ag = a;
    ar.lds_cols = MAC_RING_COLS;
    ag.ring_strips = no_ring ? 0 : MAC_RING_STRIPS;
    // (the class also takes the overflow of the shorter classes - more hits than are resident at once: those stay single-wave)
    static const int ring_min = [] {
      int L = 1;
      while (mac_length_class(L) < MAC_CLASSES - 1 && L < (1 << 20)) ++L;
      return L;
    }();
    ar.ring_min_Lt = ag.ring_min_Lt = ring_min;
    const size_t lds = mac_rows_lds(MAC_RING_COLS, false);
    // the two kinds side by side when there is a second stream and no forward list (hhv_mac_fwdlist_kernel runs over the whole class
    // between the forward and the backward kernels of both kinds)
    hipStream_t sw = (aux && aux_join && !a.fwd_list && !no_ring) ? aux : stream;
    if (!no_ring) {
      (void)hipFuncSetAttribute((const void*)hhv_mac_forward_df_kernel<LOCAL, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((hhv_mac_forward_df_kernel<LOCAL, false, true>), dim3(n_ring), dim3(MAC_DF_THREADS), lds, stream, ar);
    }
    hipLaunchKernelGGL((hhv_mac_forward_kernel<LOCAL, false, true>), dim3(n), dim3(64), 0, sw, ag);
    if (a.fwd_list) hipLaunchKernelGGL(hhv_mac_fwdlist_kernel, dim3(n), dim3(256), 0, stream, a);
    if (!no_ring) {
      if (a.fwd_list) {
        (void)hipFuncSetAttribute((const void*)hhv_mac_backward_df_kernel<LOCAL, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((hhv_mac_backward_df_kernel<LOCAL, false, true, true>), dim3(n_ring), dim3(MAC_DFB_THREADS), lds, stream, ar);
      } else {
        (void)hipFuncSetAttribute((const void*)hhv_mac_backward_df_kernel<LOCAL, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((hhv_mac_backward_df_kernel<LOCAL, false, false, true>), dim3(n_ring), dim3(MAC_DFB_THREADS), lds, stream, ar);
      }
    }
    if (a.fwd_list) hipLaunchKernelGGL((hhv_mac_backward_kernel<LOCAL, false, true, true>), dim3(n), dim3(64), 0, sw, ag);
    else hipLaunchKernelGGL((hhv_mac_backward_kernel<LOCAL, false, true, false>), dim3(n), dim3(64), 0, sw, ag);
    if (sw != stream) {  // the DP below runs over the whole class
      (void)hipEventRecord(aux_join, sw);
      (void)hipStreamWaitEvent(stream, aux_join, 0);
    }
  }
  // maximum-accuracy DP: along anti-diagonals (hhv_mac_dp_diag_kernel) unless 0.5 * mact is not a float (its chain steps are
  // float subtractions) or HHV_MAC_DP_ROWS asks for the row-by-row kernel (measurement aid)
  static const bool dp_rows = getenv("HHV_MAC_DP_ROWS") != nullptr;
  const double half = 0.5 * (double)a.mact;
  if (!dp_rows && (double)(float)half == half) {
    const size_t lds_diag = DPD_TILES + (size_t)(max_Lt + 2) * sizeof(float);
    if (lds_diag <= MAC_LDS_LIMIT) {  // (also the class without LDS for forward / backward: Sb of ~27 000 columns fits)
      (void)hipFuncSetAttribute((const void*)hhv_mac_dp_diag_kernel<LOCAL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_diag);
      hipLaunchKernelGGL((hhv_mac_dp_diag_kernel<LOCAL, false>), dim3(n), dim3(64), lds_diag, stream, a);
    } else {
      (void)hipFuncSetAttribute((const void*)hhv_mac_dp_diag_kernel<LOCAL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DPD_TILES);
      hipLaunchKernelGGL((hhv_mac_dp_diag_kernel<LOCAL, true>), dim3(n), dim3(64), DPD_TILES, stream, a);
    }
    return;
  }
  const size_t lds_dp = (size_t)2 * (max_Lt + 2) * sizeof(float);
  if (lds_dp > MAC_LDS_LIMIT) {
    hipLaunchKernelGGL((hhv_mac_dp_kernel<LOCAL, true, 4>), dim3(n), dim3(64), 0, stream, a);  // (class 6 only: row_scratch is there)
  } else if (max_Lt <= 384) {
    (void)hipFuncSetAttribute((const void*)hhv_mac_dp_kernel<LOCAL, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dp);
    hipLaunchKernelGGL((hhv_mac_dp_kernel<LOCAL, false, 1>), dim3(n), dim3(64), lds_dp, stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)hhv_mac_dp_kernel<LOCAL, false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dp);
    hipLaunchKernelGGL((hhv_mac_dp_kernel<LOCAL, false, 4>), dim3(n), dim3(64), lds_dp, stream, a);
  }
}

// Which hits of a batch the dataflow kernels take (hhv_api_mac.cpp): a workgroup of theirs is eight wavefronts at ~100 VGPRs - two per
// CU - with 112 B of LDS per template column of its class's longest template, and a batch that does not fit the GPU at once would run
// in rounds of latency-bound workgroups.  The hits beyond the budget - the shortest ones - go to the class without LDS, whose
// single-wave kernels run at up to sixteen hits per CU: slower per hit, more hits per second (2 000 hits of 300 columns: 8.6 ms
// all single-wave; 500 hits 3.3 ms dataflow against 4.0 ms single-wave).
void mac_dataflow_budget(int num_cus, int* max_hits, size_t* max_lds) {
  static const int env_hits = [] { const char* e = getenv("HHV_MAC_DF_HITS"); return e ? atoi(e) : -1; }();  // measurement aid
  *max_hits = env_hits >= 0 ? env_hits : 2 * num_cus;
  *max_lds = (size_t)num_cus * MAC_LDS_LIMIT / 10 * 9;  // (the CUs' LDS is not filled to the last byte by workgroups of mixed sizes)
}
int mac_length_class(int Lt, bool lds_allowed) {
  static const bool no_lds = getenv("HHV_MAC_NO_LDS") != nullptr;  // measurement aid: row state in global memory for every length
  // (templates beyond the dataflow kernels' LDS layout - ~1450 columns - go straight to the class without LDS: there the ring
  // variants of the dataflow kernels take the hits whose rows fit their ring, the single-wave kernels the others)
  if (lds_allowed && !no_lds && mac_rows_lds(Lt, false) <= MAC_LDS_LIMIT && (Lt + 63) / 64 <= MAC_DF_STRIPS)
    return Lt <= 128 ? 0 : Lt <= 256 ? 1 : Lt <= 384 ? 2 : Lt <= 640 ? 3 : Lt <= 1022 ? 4 : 5;
  return 6;
}

// One launch sequence (forward, backward, DP) per non-empty class.  The first non-empty class runs on the caller's stream,
// the others on side streams that wait for what the caller's stream has queued so far (inputs, masks) and are joined
// before the backtrace of all hits.
int launch_mac(const MacArgs& a0, bool local, const MacClasses& cls, void* stream_, const MacStreams* side) {
  hipStream_t stream = (hipStream_t)stream_;
  int non_empty = 0, first_of[MAC_CLASSES], at = 0;
  for (int c = 0; c < MAC_CLASSES; ++c) {
    non_empty += cls.n[c] > 0;
    first_of[c] = at;  // MacArgs::sel lists class 0 first
    at += cls.n[c];
  }
  // The classes run side by side in at most MAC_CHAINS streams: the runtime maps streams onto four hardware queues, and two classes
  // that share one run one after the other (seven streams: the longest class's forward - backward - DP chain waited behind another
  // class's, 500 mixed-length hits 10.0 -> 14.8 ms).  Classes of the longest templates first, each to the chain with the least
  // work so far (cost ~ a hit's rows x columns; the hits of a class run concurrently).
  const bool fork = side && non_empty > 1 && side->s[1];
  // stream 0 of the side streams: the single-wave kernels of the longest class (chain 0 is the caller's stream itself)
  hipStream_t aux = (side && side->s[0] && cls.n[MAC_CLASSES - 1] > 0) ? (hipStream_t)side->s[0] : nullptr;
  if (aux && !fork) {
    (void)hipEventRecord((hipEvent_t)side->fork, stream);
    (void)hipStreamWaitEvent(aux, (hipEvent_t)side->fork, 0);
  }
  static const int n_chains = [] { const char* e = getenv("HHV_MAC_CHAINS"); const int v = e ? atoi(e) : MAC_CHAINS; return v < 1 ? 1 : (v > MAC_CLASSES ? MAC_CLASSES : v); }();
  int chain_of[MAC_CLASSES];
  double load[MAC_CLASSES] = {};
  for (int c = MAC_CLASSES - 1; c >= 0; --c) {
    chain_of[c] = 0;
    if (cls.n[c] == 0 || !fork) continue;
    int best = 0;
    for (int q = 1; q < n_chains; ++q)
      if (load[q] < load[best]) best = q;
    chain_of[c] = best;
    load[best] += 256.0 + cls.max_Lt[c];
  }
  if (fork) {
    (void)hipEventRecord((hipEvent_t)side->fork, stream);  // everything queued so far: inputs, masks
    if (aux) (void)hipStreamWaitEvent(aux, (hipEvent_t)side->fork, 0);
  }
  bool used[MAC_CLASSES] = {};
  for (int c = MAC_CLASSES - 1; c >= 0; --c) {
    if (cls.n[c] == 0) continue;
    hipStream_t st = stream;
    if (fork && chain_of[c] > 0) {  // chain 0 is the caller's stream itself (it gets the longest class)
      st = (hipStream_t)side->s[chain_of[c]];
      if (!used[chain_of[c]]) (void)hipStreamWaitEvent(st, (hipEvent_t)side->fork, 0);
      used[chain_of[c]] = true;
    }
    MacArgs a = a0;
    a.sel = a0.sel + first_of[c];
    a.lds_cols = cls.max_Lt[c];
    hipStream_t ax = c == MAC_CLASSES - 1 ? aux : nullptr;
    if (local) launch_mac_class<true>(a, c, cls.n[c], cls.max_Lt[c], st, ax, ax ? (hipEvent_t)side->join[0] : nullptr, cls.n_long);
    else launch_mac_class<false>(a, c, cls.n[c], cls.max_Lt[c], st, ax, ax ? (hipEvent_t)side->join[0] : nullptr, cls.n_long);
  }
  for (int q = 0; q < MAC_CLASSES; ++q)
    if (used[q]) {
      (void)hipEventRecord((hipEvent_t)side->join[q], (hipStream_t)side->s[q]);
      (void)hipStreamWaitEvent(stream, (hipEvent_t)side->join[q], 0);
    }
  hipLaunchKernelGGL(hhv_mac_trace_kernel, dim3(a0.n), dim3(64), 0, stream, a0);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv
