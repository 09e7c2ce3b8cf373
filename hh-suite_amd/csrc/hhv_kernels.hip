// hhv_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).  No MFMA: the hot path is
// a max-plus recurrence plus a 20-term fp32 dot product whose rounding order is part of the contract.
//
// Kernel 1  hhv_stream_kernel<R, LOCAL, BT, CELLOFF>   the Viterbi DP (replaces Viterbi::Align,
//           src/hhviterbialgorithm.cpp:29-497): one 64-lane wavefront = one systolic array, see
//           viterbi_lane.h.  The wave's template stream is staged through a 14 KiB LDS ring with
//           global_load_lds_dwordx4 (HBM -> LDS without touching VGPRs), lanes read their record
//           with 7 conflict-free ds_read_b128 (28-dword stride = 16 distinct 4-bank slots), the
//           lane-to-lane hand-off is 7 v_mov_b32_dpp wave_shr:1 per step.
// Kernel 2a hhv_trace_kernel   Viterbi::Backtrace (src/hhviterbi.cpp:83-160), one lane per template
//           (serial pointer chase, O(Lq+Lt) dependent byte loads).
// Kernel 2b hhv_rescore_kernel Viterbi::ScoreForBacktrace (src/hhviterbi.cpp:195-281), one wave per template.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (fp contraction would change results).
#include <hip/hip_runtime.h>

#include "hhv_internal.h"
#include "viterbi_lane.h"

namespace hhv {

__device__ __forceinline__ float dpp_shr1(float old, float src) {
  // v_mov_b32_dpp wave_shr:1 : lane n <- lane n-1, lane 0 keeps `old`
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x138, 0xF, 0xF,
                                         false));
}
__device__ __forceinline__ int dpp_shr1(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, false);
}

// one 32-record chunk (3584 B): 3 wave-wide 16-byte-per-lane loads + one half-wave load, straight into LDS
__device__ __forceinline__ void load_chunk(const float4* __restrict__ src, int chunk, float4* ring, int lane) {
  constexpr int CHUNK_F4 = CHUNK_RECS * 7;  // 224 float4
  const float4* g = src + (size_t)chunk * CHUNK_F4 + lane;
  float4* l = ring + (chunk & (RING_CHUNKS - 1)) * CHUNK_F4;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + k * LANES),
                                     (__attribute__((address_space(3))) void*)(l + k * LANES), 16, 0, 0);
  }
  if (lane < 32) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 3 * LANES),
                                     (__attribute__((address_space(3))) void*)(l + 3 * LANES), 16, 0, 0);
  }
}

// ---- LDS operand source of a column step -----------------------------------------------------------------------------
// The ring is filled by LDS-DMA (global_load_lds).  hipcc cannot tell which ds_read may alias a DMA in flight, so in
// front of EVERY ds_read it can see it waits vmcnt(0) - which in the backtrace variants is the previous step's
// global_store of the compare bits (stores count on vmcnt): a full store round trip per step.  The loop therefore reads
// LDS only through inline asm, which hipcc does not count, and places the waits itself:
//   head()          ds_read_b128 of record dwords 20..27 (7 transitions + meta) [+ QL: the phase-A query transitions]
//                   and lgkmcnt(0) in ONE statement: everything it returns has landed.
//   begin_column()  issues the five reads of the profile values; they land under phase A.  Outputs of an asm load count
//                   as written at the end of the statement, so the values are only touched through before_B(), whose
//                   wait statement names every destination "+v" (no consumer can be scheduled above it).
//   before_B()      lgkmcnt(0) for the profile, then [QL] issues the phase-C query transitions, waited in before_C().
// Every read that is issued is waited for on the same control path, so no destination register is ever dead with a read
// still in flight.  The DMA itself is ordered by the explicit vmcnt(0) at each chunk boundary (below).
// tools/audit_asm.py checks in the generated .s that nothing touches a destination between its load and its wait.
typedef float v4f __attribute__((ext_vector_type(4)));

template <int R, bool QL_>
struct LdsColumn {
  static constexpr bool QL = QL_;
  static constexpr int NA = (R + 1) / 2;      // float4 reads covering the A block {m2i, i2i} x R (floats 0 .. 2R-1)
  static constexpr int C0 = (2 * R) / 4;      // first float4 of the C block {m2d, d2d} x R (floats 2R .. 4R-1)
  static constexpr int NC = R - C0;
  v4f v0, v1, v2, v3, v4, v5, v6;
  v4f qa0, qa1, qa2, qc0, qc1, qc2;
  uint32_t rec_addr, ql_addr;  // LDS byte addresses: this lane's record in the ring / its 20 floats of query transitions

  __device__ __forceinline__ void head() {
    if (!QL) {
      asm volatile("ds_read_b128 %0, %2 offset:96\n\tds_read_b128 %1, %2 offset:80\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v6), "=&v"(v5) : "v"(rec_addr) : "memory");
    } else if (NA == 1) {
      asm volatile("ds_read_b128 %0, %3 offset:96\n\tds_read_b128 %1, %3 offset:80\n\tds_read_b128 %2, %4\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v6), "=&v"(v5), "=&v"(qa0) : "v"(rec_addr), "v"(ql_addr) : "memory");
    } else if (NA == 2) {
      asm volatile("ds_read_b128 %0, %4 offset:96\n\tds_read_b128 %1, %4 offset:80\n\tds_read_b128 %2, %5\n\t"
                   "ds_read_b128 %3, %5 offset:16\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v6), "=&v"(v5), "=&v"(qa0), "=&v"(qa1) : "v"(rec_addr), "v"(ql_addr) : "memory");
    } else {
      asm volatile("ds_read_b128 %0, %5 offset:96\n\tds_read_b128 %1, %5 offset:80\n\tds_read_b128 %2, %6\n\t"
                   "ds_read_b128 %3, %6 offset:16\n\tds_read_b128 %4, %6 offset:32\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(v6), "=&v"(v5), "=&v"(qa0), "=&v"(qa1), "=&v"(qa2) : "v"(rec_addr), "v"(ql_addr) : "memory");
    }
  }
  __device__ __forceinline__ int32_t meta() const {
    const float w = v6.w;  // (bit_cast applied to the element expression itself reads element 0 with this clang)
    return __builtin_bit_cast(int32_t, w);
  }
  // header record: dword 0 = template index (read and waited for in one statement)
  __device__ __forceinline__ int32_t header_tid() const {
    int32_t t;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(t) : "v"(rec_addr));
    return t;
  }
  __device__ __forceinline__ void begin_column() {
    asm volatile("ds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b128 %2, %5 offset:32\n\t"
                 "ds_read_b128 %3, %5 offset:48\n\tds_read_b128 %4, %5 offset:64"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4) : "v"(rec_addr));
  }
  __device__ __forceinline__ float tr(int k) const {
    switch (k) {
      case 0: return v5.x;
      case 1: return v5.y;
      case 2: return v5.z;
      case 3: return v5.w;
      case 4: return v6.x;
      case 5: return v6.y;
      default: return v6.z;
    }
  }
  __device__ __forceinline__ void before_B() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4));
    if (QL) {
      // float4 C0 .. R-1 of the lane's 20 floats (for odd R the first one straddles the A block and is read again)
      if (NC == 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(qc0) : "v"(ql_addr), "i"(16 * C0));
      if (NC == 2)
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                     : "=&v"(qc0), "=&v"(qc1) : "v"(ql_addr), "i"(16 * C0), "i"(16 * C0 + 16));
      if (NC == 3)
        asm volatile("ds_read_b128 %0, %3 offset:%4\n\tds_read_b128 %1, %3 offset:%5\n\tds_read_b128 %2, %3 offset:%6"
                     : "=&v"(qc0), "=&v"(qc1), "=&v"(qc2) : "v"(ql_addr), "i"(16 * C0), "i"(16 * C0 + 16), "i"(16 * C0 + 32));
    }
  }
  __device__ __forceinline__ void get_p(float* tp) const {
    tp[0] = v0.x, tp[1] = v0.y, tp[2] = v0.z, tp[3] = v0.w;
    tp[4] = v1.x, tp[5] = v1.y, tp[6] = v1.z, tp[7] = v1.w;
    tp[8] = v2.x, tp[9] = v2.y, tp[10] = v2.z, tp[11] = v2.w;
    tp[12] = v3.x, tp[13] = v3.y, tp[14] = v3.z, tp[15] = v3.w;
    tp[16] = v4.x, tp[17] = v4.y, tp[18] = v4.z, tp[19] = v4.w;
  }
  __device__ __forceinline__ void before_C() {
    if (QL) {
      if (NC == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qc0));
      if (NC == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qc0), "+v"(qc1));
      if (NC == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qc0), "+v"(qc1), "+v"(qc2));
    }
  }
  static __device__ __forceinline__ float pick(const v4f& a, const v4f& b, const v4f& c, int f) {
    const v4f& v = f < 4 ? a : (f < 8 ? b : c);
    switch (f & 3) {
      case 0: return v.x;
      case 1: return v.y;
      case 2: return v.z;
      default: return v.w;
    }
  }
  // lane layout in LDS: floats [0, 2R) = {m2i, i2i} of rows 0..R-1, floats [2R, 4R) = {m2d, d2d}
  __device__ __forceinline__ float qa(int r, int w) const { return pick(qa0, qa1, qa2, 2 * r + w); }
  __device__ __forceinline__ float qc(int r, int w) const { return pick(qc0, qc1, qc2, 2 * R + 2 * r + w - 4 * C0); }
};

// Occupancy: VALU issue needs >= 2 waves per SIMD to reach its rate on gfx950 (a lone wave issues one
// VOP2 every ~6.3 clk, two waves one every ~2.5 clk: tools/valu_ubench).  Up to R = 5 rows per lane the
// kernel is held to <= 256 VGPRs (2 waves/SIMD, no scratch in any variant).
// MULTI = the query needs more than one pass of 64*R rows (the carry hand-over code is compiled out otherwise).
// SS = secondary-structure term added to the emission score (the reference's ...AndSS builds).
template <int R, bool LOCAL, bool BT, bool CELLOFF, bool MULTI, bool SS>
__global__ void __launch_bounds__(LANES, 2) hhv_stream_kernel(StreamArgs a) {
  // backtrace variants: {m2i, i2i} and {m2d, d2d} of the lane's R query rows live in LDS (20 floats per lane; the
  // 20-dword stride is conflict-free for ds_read_b128) - the VGPRs they would occupy hold the compare results instead.
  // ONE __shared__ object: [QL block][ring].
  constexpr bool QL = BT;
  constexpr int QL_F4 = QL ? LANES * 5 : 0;
  __shared__ float4 smem[QL_F4 + RING_RECS * 7];
  float4* const ring = smem + QL_F4;
  const int lane = threadIdx.x;
  const int64_t rb = a.wave_rec[blockIdx.x];
  const int64_t re = a.wave_rec[blockIdx.x + 1];
  if (re <= rb) return;
  const int M = (int)(re - rb) + 1;  // the range's records plus the next header (finalizes the last template)

  Params P;
  P.egq = a.egq;
  P.egt = a.egt;
  P.shift = a.shift;
  P.Lq = a.Lq;
  const int i0 = a.row_base + lane * R + 1;
  // the lane that emits results: owner of row Lq in the last pass, lane 63 otherwise
  const int g_last = (!MULTI || a.pass_last) ? (a.Lq - a.row_base - 1) / R : LANES - 1;
  const int r_last = (a.Lq - a.row_base - 1) % R;
  const bool first = !MULTI || a.pass_first != 0;
  const bool carry_out = MULTI && a.pass_last == 0;

  const float4* src = (const float4*)a.records + rb * 7;
  const int nchunks = (M + CHUNK_RECS - 1) / CHUNK_RECS;
  load_chunk(src, 0, ring, lane);
  load_chunk(src, 1, ring, lane);  // the stream is padded: over-reading past M is harmless
  static_assert((RING_RECS & (RING_RECS - 1)) == 0 && RING_CHUNKS == 4 && CHUNK_RECS == 32, "ring geometry");

  QRows<R> q;
  q.load(a.qpack + (size_t)lane * R * REC_DW);
  LdsColumn<R, QL> col;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const uint32_t ring_addr = smem_addr + QL_F4 * 16;
  col.ql_addr = smem_addr + lane * 80;
  if (QL) {
    float* w = reinterpret_cast<float*>(smem) + lane * 20;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      w[2 * r + 0] = q.m2i[r];
      w[2 * r + 1] = q.i2i[r];
      w[2 * R + 2 * r + 0] = q.m2d[r];
      w[2 * R + 2 * r + 1] = q.d2d[r];
    }
  }
  LaneState<R> st;
  st.reset();
  int ss_qoff[R];
  if (SS) {
#pragma unroll
    for (int r = 0; r < R; ++r) ss_qoff[r] = a.ss_q_off[i0 - 1 + r];
  }

  // chunks 0 and 1 have landed, the lane's own ds_writes above are done (LDS executes a wave's operations in order)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  for (int s = 0; s < M + LANES - 1; ++s) {
    if ((s & (CHUNK_RECS - 1)) == 0 && s > 0) {
      // chunk c = s/32 was issued 32 steps ago: make sure it has landed, then refill the slot that
      // held chunk c-3 (its last reader, lane 63, finished at step 32(c-2)+62 < 32c) with chunk c+1.
      // The live window [s-63, s] spans chunks c-2..c, so the ring holds 4 chunks = 128 records.
      // (The same wait retires the backtrace stores of the last 32 steps - the only place they are waited for.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int c = s / CHUNK_RECS;
      if (c + 1 < nchunks) load_chunk(src, c + 1, ring, lane);
    }
    const int r = s - lane;
    const bool active = (r >= 0) && (r < M);

    col.rec_addr = ring_addr + (uint32_t)((s - lane) & (RING_RECS - 1)) * (REC_DW * 4);
    col.head();
    const int32_t meta = col.meta();

    // hand-off from lane g-1 (full EXEC here); lane 0 takes the DP boundary row 0, or - in later passes of
    // a long query - the bottom row the previous pass left for this record (and its running best)
    Incoming bnd = boundary_incoming(meta, P);
    if (!first) {
      if (lane == 0 && active) {
        const float4 c = a.carry[rb + r];
        bnd.MM = c.x;
        bnd.GD = c.y;
        bnd.IM = c.z;
        bnd.DG = c.w;
        bnd.MI = a.carry_mi[rb + r];
        if (meta < 0 && st.tid >= 0) {
          const DevResult pr = a.results[st.tid & TID_MASK];
          bnd.fs = pr.score;
          bnd.fpos = (pr.i2 << 16) | pr.j2;
        }
      }
    }
    Incoming in;
    in.MM = dpp_shr1(bnd.MM, st.MM[R - 1]);
    in.GD = dpp_shr1(bnd.GD, st.GD[R - 1]);
    in.IM = dpp_shr1(bnd.IM, st.IM[R - 1]);
    in.DG = dpp_shr1(bnd.DG, st.DG[R - 1]);
    in.MI = dpp_shr1(bnd.MI, st.MI[R - 1]);
    in.fs = dpp_shr1(bnd.fs, st.fs);
    in.fpos = dpp_shr1(bnd.fpos, st.fpos);

    if (active) {
      if (meta < 0) {
        TemplateResult res;
        const int new_tid = col.header_tid() | ((meta & META_NOLASTCOL) ? TID_NOLASTCOL : 0);
        if (lane_header<R, LOCAL, true>(st, q, in, i0, new_tid, P, lane == g_last, res)) {
          DevResult o;
          o.score = res.score;
          o.i2 = res.i2;
          o.j2 = res.j2;
          o.index = res.tid;
          a.results[res.tid] = o;
        }
      } else {
        const int j = meta & META_JMASK;
        uint64_t cell = 0;
        uint64_t* bte = nullptr;
        if (BT || CELLOFF)
          bte = a.bt + ((MULTI ? (size_t)a.bt_plane * a.bt_pass_stride : 0) + (size_t)(rb + r) * LANES + lane);
        if (CELLOFF) cell = *bte;
        float ssv[R];
        if (SS) {
          const int tidx = (meta >> a.ss_t_shift) & a.ss_t_mask;
#pragma unroll
          for (int r = 0; r < R; ++r) ssv[r] = a.ss_table[ss_qoff[r] + tidx];
        }
        const uint64_t bytes = lane_column<R, LOCAL, BT, CELLOFF, true, SS>(st, q, in, col, j, i0, r_last, P, cell, ssv);
        if (BT) *bte = bytes;
      }
      if (carry_out) {
        if (lane == LANES - 1) {
          a.carry[rb + r] = make_float4(st.MM[R - 1], st.GD[R - 1], st.IM[R - 1], st.DG[R - 1]);
          a.carry_mi[rb + r] = st.MI[R - 1];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backtrace + rescoring, one lane per template.

// src/util-inl.h:108-130 (tables built on the host exactly like the reference builds them)
__device__ __forceinline__ float fast_log2_dev(float x, const float* __restrict__ lg2, const float* __restrict__ diff) {
  if (x <= 0) return -100000;
  const uint32_t u = f2bits(x);
  const int aa = (int)((u & 0x7F800000u) >> 23) - 0x7f;
  const int b = (int)((u & 0x007FE000u) >> 13);
  const int c = (int)(u & 0x00001FFFu);
  return ((float)aa + lg2[b]) + diff[b] * (float)c;
}

// src/hhhit-inl.h:125-131: plain left-to-right sum (the SSE branch above it is never compiled)
__device__ __forceinline__ float dot20_scalar_dev(const float* __restrict__ q, const float* __restrict__ t) {
  float r = t[0] * q[0];
#pragma unroll
  for (int k = 1; k < 20; ++k) r = r + t[k] * q[k];
  return r;
}

// Kernel 2a: Viterbi::Backtrace (src/hhviterbi.cpp:83-160) - a serial pointer chase, one lane per template.
__global__ void __launch_bounds__(64) hhv_trace_kernel(TraceArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  const DevResult res = a.results[k];
  const int64_t rec0 = a.rec_off[k];
  const int64_t po = a.path_off[k];
  int32_t* i_steps = a.i_steps + po;
  int32_t* j_steps = a.j_steps + po;
  int8_t* states = a.states + po;

  int step = 0, matched = 0;
  int i = res.i2, j = res.j2;
  int state = 2;  // MM
  while (state != 0) {
    step++;
    states[step] = (int8_t)state;
    i_steps[step] = i;
    j_steps[step] = j;
    uint32_t b = 0;
    if (i >= 1 && j >= 1) {
      int pass, g, rr, Rp;
      a.plan.locate(i, pass, g, rr, Rp);
      b = bt_decode(a.bt[(size_t)pass * a.bt_pass_stride + (size_t)(rec0 + j) * LANES + g], rr, Rp);
    }
    switch (state) {
      case 2:  // MM
        matched++;
        if (i <= 1 || j <= 1) state = 0;
        else {
          state = b & 7;
          i--;
          j--;
        }
        break;
      case 3:  // GD
        if (j <= 1) state = 0;
        else {
          if (b & 8) state = 2;
          j--;
        }
        break;
      case 4:  // IM
        if (j <= 1) state = 0;
        else {
          if (b & 16) state = 2;
          j--;
        }
        break;
      case 5:  // DG
        if (i <= 1) state = 0;
        else {
          if (b & 32) state = 2;
          i--;
        }
        break;
      case 6:  // MI
        if (i <= 1) state = 0;
        else {
          if (b & 64) state = 2;
          i--;
        }
        break;
      default:  // :139-144
        state = 0;
        break;
    }
  }
  states[step] = 2;  // :147
  DevHit h;
  h.score = res.score;  // completed by hhv_rescore_kernel
  h.viterbi_score = res.score;
  h.score_ss = 0.0f;
  h.index = k;
  h.i1 = i_steps[step];
  h.j1 = j_steps[step];
  h.i2 = res.i2;
  h.j2 = res.j2;
  h.nsteps = step;
  h.matched_cols = matched;
  a.hits[k] = h;
}

// Kernel 2b: Viterbi::ScoreForBacktrace (src/hhviterbi.cpp:195-281), one wavefront per template: the per-step
// column scores S are independent (lanes stride over the steps); the four correlation sums are then
// accumulated by one lane in exactly the reference's order (:241-249), reading S from LDS.
constexpr int RESCORE_LDS_FLOATS = 4096;
__global__ void __launch_bounds__(64) hhv_rescore_kernel(TraceArgs a) {
  __shared__ float sS[RESCORE_LDS_FLOATS];
  const int k = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t rec0 = a.rec_off[k];
  const int64_t po = a.path_off[k];
  const int32_t* i_steps = a.i_steps + po;
  const int32_t* j_steps = a.j_steps + po;
  const int8_t* states = a.states + po;
  float* S = a.S + po;
  const int nsteps = a.hits[k].nsteps;
  const bool in_lds = nsteps + 1 <= RESCORE_LDS_FLOATS;
  for (int s = 1 + lane; s <= nsteps; s += LANES) {
    float v = 0.0f;
    if (states[s] == 2) {
      const float* qp = a.qp + (size_t)i_steps[s] * 20;
      const float* tp = a.records + (size_t)(rec0 + j_steps[s]) * REC_DW;
      v = fast_log2_dev(dot20_scalar_dev(qp, tp), a.lg2, a.diff);
    }
    S[s] = v;
    if (in_lds) sS[s] = v;
  }
  __syncthreads();
  if (lane == 0) {
    float score = a.hits[k].viterbi_score;
    // :225-238: score_ss = sum over MM steps of ScoreSS(q,t,i,j) in step order; subtracted when ssm == 2
    float score_ss = 0.0f;
    if (a.ss_table) {
      for (int s = 1; s <= nsteps; ++s) {
        if (states[s] == 2 && i_steps[s] >= 1 && j_steps[s] >= 1) {
          const int32_t meta = __builtin_bit_cast(int32_t, a.records[(size_t)(rec0 + j_steps[s]) * REC_DW + REC_META]);
          score_ss += a.ss_table[a.ss_q_off[i_steps[s] - 1] + ((meta >> a.ss_t_shift) & a.ss_t_mask)];
        }
      }
    }
    if (a.ss_mode == 2) score -= score_ss;
    a.hits[k].score_ss = score_ss;
    float Scorr = 0;
    if (in_lds) {
      for (int s = 2; s <= nsteps; ++s) Scorr += sS[s] * sS[s - 1];
      for (int s = 3; s <= nsteps; ++s) Scorr += sS[s] * sS[s - 2];
      for (int s = 4; s <= nsteps; ++s) Scorr += sS[s] * sS[s - 3];
      for (int s = 5; s <= nsteps; ++s) Scorr += sS[s] * sS[s - 4];
    } else {
      __threadfence_block();
      for (int s = 2; s <= nsteps; ++s) Scorr += S[s] * S[s - 1];
      for (int s = 3; s <= nsteps; ++s) Scorr += S[s] * S[s - 2];
      for (int s = 4; s <= nsteps; ++s) Scorr += S[s] * S[s - 3];
      for (int s = 5; s <= nsteps; ++s) Scorr += S[s] * S[s - 4];
    }
    score += a.corr * Scorr;
    a.hits[k].score = score;
  }
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers

template <int R, bool LOCAL, bool BT, bool CELLOFF>
static void* kernel_ptr(bool multi, bool ss) {
  if (ss)
    return multi ? (void*)hhv_stream_kernel<R, LOCAL, BT, CELLOFF, true, true>
                 : (void*)hhv_stream_kernel<R, LOCAL, BT, CELLOFF, false, true>;
  return multi ? (void*)hhv_stream_kernel<R, LOCAL, BT, CELLOFF, true, false>
               : (void*)hhv_stream_kernel<R, LOCAL, BT, CELLOFF, false, false>;
}

template <int R>
static void* pick_variant(bool local, bool bt, bool celloff, bool multi, bool ss) {
  if (celloff) return local ? kernel_ptr<R, true, true, true>(multi, ss) : kernel_ptr<R, false, true, true>(multi, ss);
  if (bt) return local ? kernel_ptr<R, true, true, false>(multi, ss) : kernel_ptr<R, false, true, false>(multi, ss);
  return local ? kernel_ptr<R, true, false, false>(multi, ss) : kernel_ptr<R, false, false, false>(multi, ss);
}

static void* pick(int R, bool local, bool bt, bool celloff, bool multi, bool ss) {
  switch (R) {
    case 1: return pick_variant<1>(local, bt, celloff, multi, ss);
    case 2: return pick_variant<2>(local, bt, celloff, multi, ss);
    case 3: return pick_variant<3>(local, bt, celloff, multi, ss);
    case 4: return pick_variant<4>(local, bt, celloff, multi, ss);
    case 5: return pick_variant<5>(local, bt, celloff, multi, ss);
  }
  return nullptr;
}

int launch_stream(int R, bool local, bool bt, bool celloff, bool multi, bool ss, const StreamArgs& a, int n_waves,
                  void* stream) {
  void* fn = pick(R, local, bt, celloff, multi, ss);
  if (!fn) return -1;
  StreamArgs args = a;
  void* kargs[] = {&args};
  hipError_t e = hipLaunchKernel(fn, dim3(n_waves), dim3(LANES), kargs, 0, (hipStream_t)stream);
  return e == hipSuccess ? 0 : -(int)e;
}

int stream_kernel_occupancy(int R, bool local, bool bt, bool celloff, bool multi, bool ss, int* blocks_per_cu,
                            int* vgprs) {
  void* fn = pick(R, local, bt, celloff, multi, ss);
  if (!fn) return -1;
  int nb = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, LANES, 0);
  if (e != hipSuccess) return -(int)e;
  hipFuncAttributes fa;
  e = hipFuncGetAttributes(&fa, fn);
  if (e != hipSuccess) return -(int)e;
  if (blocks_per_cu) *blocks_per_cu = nb;
  if (vgprs) *vgprs = fa.numRegs;
  return 0;
}

int launch_trace(const TraceArgs& a, void* stream) {
  // 64 templates per wave for the chase (latency bound: many small blocks spread over all CUs)
  hipLaunchKernelGGL(hhv_trace_kernel, dim3((a.n + LANES - 1) / LANES), dim3(LANES), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(hhv_rescore_kernel, dim3(a.n), dim3(LANES), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv
