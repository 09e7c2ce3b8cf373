// hhv_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4).  No MFMA: the hot path is
// a max-plus recurrence plus a 20-term fp32 dot product whose rounding order is part of the contract.
//
// Kernel 1  hhv_stream_kernel<R, LOCAL, BT, CELLOFF, MULTI, SS, W> (hhv_stream_kernel.h)   the Viterbi DP (replaces
//           Viterbi::Align, src/hhviterbialgorithm.cpp:29-497): one 64-lane wavefront = one systolic array (W = 64;
//           this unit) or two / four arrays for short queries (hhv_kernels_w32.hip / _w16.hip), see viterbi_lane.h.  The wave's template stream is staged through a 14 KiB LDS ring with
//           global_load_lds_dwordx4 (HBM -> LDS without touching VGPRs), lanes read their record
//           with 7 conflict-free ds_read_b128 (28-dword stride = 16 distinct 4-bank slots), the
//           lane-to-lane hand-off is 5 ds_bpermute_b32 per step (+ 2 in steps with a header), delivered late (hhv_stream_kernel.h).
// Kernel 2a hhv_trace_kernel   Viterbi::Backtrace (src/hhviterbi.cpp:83-160), one lane per template
//           (serial pointer chase, O(Lq+Lt) dependent byte loads).
// Kernel 2b hhv_rescore_kernel Viterbi::ScoreForBacktrace (src/hhviterbi.cpp:195-281), per-step scores: one wave per template;
// Kernel 2c hhv_scorr_kernel   its correlation sums and the Hit score: one lane per template.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (fp contraction would change results).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "hhv_internal.h"
#include "hhv_stream_kernel.h"
#include "viterbi_lane.h"

#if defined(HHV_EXP_TIMING)
extern "C" __attribute__((visibility("default"))) int hhv_debug_clk(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hhv::hhv_dbg_clk), 8 * sizeof(unsigned long long));
}
#endif

#if defined(HHV_EXP_WAVETIME)
extern "C" __attribute__((visibility("default"))) int hhv_debug_wave(unsigned long long* out, int n_waves) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(hhv::hhv_dbg_wave), (size_t)n_waves * 4 * sizeof(unsigned long long));
}
#endif

namespace hhv {

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

// one step of Viterbi::Backtrace (src/hhviterbi.cpp:96-146); b = the reference's backtrace byte of cell (i, j)
__device__ __forceinline__ void trace_step(int& state, int& i, int& j, int& matched, uint32_t b, uint32_t* err) {
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
    default:  // :139-144: the reference reports "unallowed state value" and ends the path; here the context's error word
      if (err) __hip_atomic_fetch_or(err, DEV_ERR_TRACE_STATE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      state = 0;
      break;
  }
}

// Kernel 2a: Viterbi::Backtrace (src/hhviterbi.cpp:83-160) - a serial pointer chase, one lane per template.
// Every step reads one 8-byte entry that depends on the step before - an HBM round trip per step if done naively
// (the entries of a launch are written once, gigabytes ago).  Single-pass plans therefore read a WINDOW per round trip:
// from cell (i, j) with entry (row, g) a path can only move to rows row, row - 1, row - 2 and to lane g or g - 1
// (bt_entry: row = record + lane; a step lowers i and / or j by one), so the ten entries {row .. row - 4} x {g - 1, g} -
// five independent 16-byte reads - cover the next two steps at least, four on a diagonal; the walk continues out of
// registers until it leaves the window.
// What the walk WRITES is one byte per step, the state (round 4): with large sets the kernel is bound by memory
// transactions - 64 lanes on 64 different paths, every access its own 64-byte sector - and i_steps / j_steps (two scattered
// 4-byte stores per step and lane) follow from the states and the end point: every recorded step but the last lowers i
// (MM, DG, MI) and / or j (MM, GD, IM) by one.  hhv_rescore_kernel, which needs (i, j) of every step anyway, rebuilds them
// with two ballots per 64 steps and writes them as contiguous 256-byte rows.  The state bytes leave in words of four (pools
// start on multiples of four).
// With small sets (fewer wavefronts than SIMDs) the walk's own instructions are the time: RC / MMC = rows per lane and
// encoding of the MM predecessor as compile-time constants for the single-pass plans (the row -> lane division and the
// decoder's shifts fold), 0 = run-time values (multi-pass plans).  (Requesting the window a diagonal walk needs next together
// with the current one - the paths of the benchmark are 96 % diagonal - made the kernel slower at 10 k templates as well as
// at 100 k, 253 -> 305 us and 608 -> 640 us: the loads were not what a round waits for.  profiles/r4_ab.txt.)
template <int RC, int MMC>
__global__ void __launch_bounds__(64) hhv_trace_kernel(TraceArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.n) return;
  const DevResult res = a.results[k];
  const int64_t rec0 = a.rec_off[k];
  const int64_t po = a.path_off[k];
  uint32_t* state_words = reinterpret_cast<uint32_t*>(a.states + po);  // (po % 4 == 0: ensure_paths)

  int step = 0, matched = 0;
  int i = res.i2, j = res.j2;
  int last_i = i, last_j = j;
  int state = 2;  // MM
  uint32_t sbuf = 0;  // states of steps 4 w .. 4 w + 3, byte (step & 3) each; byte 0 of word 0 = the unused index 0
  // one step of the walk: the state recorded for it is known after the step (the LAST step is recorded as MM, :147)
  auto walk = [&](uint32_t b) __attribute__((always_inline)) {
    const int st_here = state;
    last_i = i;
    last_j = j;
    trace_step(state, i, j, matched, b, a.err);
    step++;
    sbuf |= (uint32_t)(state == 0 ? 2 : st_here) << (8 * (step & 3));
    if ((step & 3) == 3) {
      state_words[step >> 2] = sbuf;
      sbuf = 0;
    }
  };
  if (RC > 0) {
    constexpr int R = RC > 0 ? RC : 1;
    const int W = a.plan.W;
    constexpr int WIN = 5;
    while (state != 0) {
      const int g0 = i >= 1 ? (i - 1) / R : 0;
      const int c0 = max(g0 - 1, 0);       // first of the two columns of the window (c0 + 1 <= W - 1)
      const int64_t row0 = rec0 + j + g0;  // row of the entry of (i, j)
      uint64_t w[WIN][2];
#pragma unroll
      for (int d = 0; d < WIN; ++d) {
        const uint64_t* e = a.bt + (size_t)max<int64_t>(row0 - d, 0) * (size_t)W + (size_t)c0;
        w[d][0] = e[0];
        w[d][1] = e[1];
      }
#pragma unroll
      for (int sub = 0; sub < WIN + 1; ++sub) {
        if (state == 0) break;
        uint32_t b = 0;
        if (i >= 1 && j >= 1) {
          const int g = (i - 1) / R, rr = (i - 1) - g * R;
          const int d = (int)(row0 - (rec0 + j + g));  // rows behind the anchor, >= 0
          const int c = g - c0;                        // 0 or 1 inside the window
          if (d >= WIN || c < 0) break;                // left the window: next round trip (at least one step was made)
          uint64_t entry = c ? w[0][1] : w[0][0];
#pragma unroll
          for (int t = 1; t < WIN; ++t) entry = (d == t) ? (c ? w[t][1] : w[t][0]) : entry;
          b = bt_decode(entry, rr, R, MMC);
        }
        walk(b);
      }
    }
  } else {
    while (state != 0) {
      uint32_t b = 0;
      if (i >= 1 && j >= 1) {
        int pass, g, rr, Rp;
        a.plan.locate(i, pass, g, rr, Rp);
        b = bt_decode(a.bt[(size_t)pass * a.bt_pass_stride + bt_entry(rec0 + j, g, a.plan.W)], rr, Rp, a.bt_mm);
      }
      walk(b);
    }
  }
  if ((step & 3) != 3) state_words[step >> 2] = sbuf;  // the last, partial word (its upper bytes lie inside the template's own pool)
  DevHit h;
  h.score = res.score;  // completed by hhv_scorr_kernel
  h.viterbi_score = res.score;
  h.score_ss = 0.0f;
  h.index = k;
  h.i1 = last_i;
  h.j1 = last_j;
  h.i2 = res.i2;
  h.j2 = res.j2;
  h.nsteps = step;
  h.matched_cols = matched;
  a.hits[k] = h;
}

// Kernel 2a', round 5: the same walk with one WAVEFRONT per template - for the set sizes of real searches (hhblits aligns <= 20 000
// templates after its prefilter), where one lane per template leaves the device nearly empty (10 000 templates = 157 wavefronts
// on 1 024 SIMDs) and the walk's ~75 dependent round trips are the kernel's time.  A path is a sequence of RUNS - steps that
// stay in one state: a diagonal stretch of match states, a gap of GD / IM (column by column) or DG / MI (row by row) - and inside
// a run every step's cell is known in advance.  So per round trip the 64 lanes load and decode the entries of the NEXT 64 cells
// of the current run's direction, a ballot finds the first lane whose entry ends the run (the predecessor code of a match
// state is not MM, the "gap closes" bit of a gap state is set, or the matrix border is reached), and lanes 0 .. n record their
// step: a round trip per RUN (a dozen per alignment) instead of one per two to four steps.  Same states, same end points,
// same counts as trace_step above (src/hhviterbi.cpp:96-146) - tests compare the two kernels byte for byte.
template <int RC, int MMC>
__global__ void __launch_bounds__(64) hhv_trace_wave_kernel(TraceArgs a) {
  const int k = blockIdx.x;
  const int lane = threadIdx.x;
  const DevResult res = a.results[k];
  const int64_t rec0 = a.rec_off[k];
  int8_t* const states = a.states + a.path_off[k];
  const int W = a.plan.W;
  if (lane == 0) states[0] = 0;  // (index 0 is unused: steps count from 1)
  int i = res.i2, j = res.j2;   // the cell of the next step (wave uniform)
  int state = 2, step = 0, matched = 0, last_i = i, last_j = j;
  while (state != 0) {
    // :139-144: an illegal state - the reference reports it, counts the step and ends the walk; here: the error word, and the
    // step runs through the code below as a run of one step that ends the walk (recorded as MM like every last step, :147)
    const bool illegal = state < 2 || state > 6;
    if (illegal && a.err && lane == 0) __hip_atomic_fetch_or(a.err, DEV_ERR_TRACE_STATE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const bool mm = state == 2, horizontal = state == 3 || state == 4;
    const int di = (mm || !horizontal) ? 1 : 0, dj = (mm || horizontal) ? 1 : 0;
    const int il = i - lane * di, jl = j - lane * dj;  // this lane's cell, if the run lasts that long
    uint32_t b = 0;
    if (il >= 1 && jl >= 1) {
      int pass = 0, g, rr, Rp = RC;
      if (RC > 0) {
        g = (il - 1) / RC;
        rr = (il - 1) - g * RC;
      } else {
        a.plan.locate(il, pass, g, rr, Rp);
      }
      b = bt_decode(a.bt[(size_t)pass * a.bt_pass_stride + bt_entry(rec0 + jl, g, W)], rr, Rp, RC > 0 ? MMC : a.bt_mm);
    }
    // does this lane's step END the run?  border: the walk stops here; otherwise the step leaves the run's state
    const bool border = mm ? (il <= 1 || jl <= 1) : horizontal ? jl <= 1 : il <= 1;
    const uint32_t bit = state == 3 ? 8u : state == 4 ? 16u : state == 5 ? 32u : 64u;
    const bool leaves = mm ? (b & 7u) != 2u : (b & bit) != 0u;
    const unsigned long long ends = __ballot(border || leaves || illegal);
    const int n = ends ? (int)__builtin_ctzll(ends) : 63;                  // the run's last step of this round
    const bool closed = ends != 0;                                         // ... really ends the run
    const int b_n = __builtin_amdgcn_readlane((int)b, n);
    const int border_n = (int)((__ballot(border) >> n) & 1ull);
    const int next = illegal ? 0 : !closed ? state : border_n ? 0 : mm ? (b_n & 7) : 2;  // state behind step n
    // steps step + 1 .. step + n + 1 are the cells of lanes 0 .. n; recorded state = the run's, the walk's LAST step as MM (:147).
    // The lanes know their step's cell, so what Viterbi::ScoreForBacktrace needs per step (src/hhviterbi.cpp:222-237) is written
    // here as well - (i, j), the column score S = fast_log2(ScalarProd20(q.p[i], t.p[j])) of the match steps, the secondary-
    // structure score - with the expressions of hhv_rescore_kernel (which the lane-per-template walk is followed by); the loads
    // are in flight together with the next run's entries.
    if (lane <= n) {
      const int st = (lane == n && next == 0) ? 2 : state;
      const int64_t o = a.path_off[k] + step + 1 + lane;
      states[step + 1 + lane] = (int8_t)st;
      a.i_steps[o] = il;
      a.j_steps[o] = jl;
      float v = 0.0f;
      if (st == 2) {
        const float4* qp = reinterpret_cast<const float4*>(a.qp + (size_t)il * 20);
        const float4* tp = reinterpret_cast<const float4*>(a.records + (size_t)(rec0 + jl) * REC_DW);
        float q[20], t[20];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
          const float4 qa = qp[x], ta = tp[x];
          q[4 * x + 0] = qa.x, q[4 * x + 1] = qa.y, q[4 * x + 2] = qa.z, q[4 * x + 3] = qa.w;
          t[4 * x + 0] = ta.x, t[4 * x + 1] = ta.y, t[4 * x + 2] = ta.z, t[4 * x + 3] = ta.w;
        }
        v = fast_log2_dev(dot20_scalar_dev(q, t), a.lg2, a.diff);
      }
      a.S[o] = v;
      if (a.Sss) {
        float vs = 0.0f;
        if (st == 2 && il >= 1 && jl >= 1) {
          const int32_t meta = __builtin_bit_cast(int32_t, a.records[(size_t)(rec0 + jl) * REC_DW + REC_META]);
          vs = a.ss_table[a.ss_q_off[il - 1] + ((meta >> a.ss_t_shift) & a.ss_t_mask)];
        }
        a.Sss[o] = vs;
      }
    }
    step += n + 1;
    if (mm) matched += n + 1;
    last_i = i - n * di;
    last_j = j - n * dj;
    // the cell of the next step: a step that reaches the border does not move (:104-135)
    const int moved = (closed && border_n) ? n : n + 1;
    i -= moved * di;
    j -= moved * dj;
    state = next;
  }
  if (lane == 0) {
    DevHit h;
    h.score = res.score;  // completed by hhv_scorr_kernel
    h.viterbi_score = res.score;
    h.score_ss = 0.0f;
    h.index = k;
    h.i1 = last_i;
    h.j1 = last_j;
    h.i2 = res.i2;
    h.j2 = res.j2;
    h.nsteps = step;
    h.matched_cols = matched;
    a.hits[k] = h;
  }
}

// Kernel 2b: Viterbi::ScoreForBacktrace (src/hhviterbi.cpp:195-281), first half - one wavefront per template: (i, j) of
// every step rebuilt from the state bytes (see hhv_trace_kernel) and written out, and the per-step column scores
// S[step] = fast_log2(ScalarProd20(q.p[i], t.p[j])) of the MM steps (:225-235), lanes striding over the steps.  The template
// columns are read with 16-byte loads straight from the record stream: this kernel reads the database a second time and
// runs at the speed that read allows.
__global__ void __launch_bounds__(64) hhv_rescore_kernel(TraceArgs a) {
  const int k = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t rec0 = a.rec_off[k];
  const int64_t po = a.path_off[k];
  int32_t* i_steps = a.i_steps + po;
  int32_t* j_steps = a.j_steps + po;
  const int8_t* states = a.states + po;
  float* S = a.S + po;
  const DevHit h = a.hits[k];
  const int nsteps = h.nsteps;
  int ci = h.i2, cj = h.j2;  // (i, j) of the first step of the chunk
  for (int s0 = 1; s0 <= nsteps; s0 += LANES) {
    const int s = s0 + lane;
    const int st = s <= nsteps ? (int)states[s] : 0;
    // step s + 1 starts at (i, j) of step s lowered by the move of state st (src/hhviterbi.cpp:96-146); the state recorded
    // for the last step (always MM, :147) is not followed by a move - no step reads it
    const bool di = st == 2 || st == 5 || st == 6, dj = st == 2 || st == 3 || st == 4;
    const unsigned long long mi = __ballot(di), mj = __ballot(dj);
    const int i = ci - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mi >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mi, 0u));
    const int j = cj - (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mj >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mj, 0u));
    ci -= (int)__popcll(mi);
    cj -= (int)__popcll(mj);
    if (s <= nsteps) {
      i_steps[s] = i;
      j_steps[s] = j;
      float v = 0.0f;
      if (st == 2) {
        const float4* qp = reinterpret_cast<const float4*>(a.qp + (size_t)i * 20);
        const float4* tp = reinterpret_cast<const float4*>(a.records + (size_t)(rec0 + j) * REC_DW);
        float q[20], t[20];
#pragma unroll
        for (int x = 0; x < 5; ++x) {
          const float4 qa = qp[x], ta = tp[x];
          q[4 * x + 0] = qa.x, q[4 * x + 1] = qa.y, q[4 * x + 2] = qa.z, q[4 * x + 3] = qa.w;
          t[4 * x + 0] = ta.x, t[4 * x + 1] = ta.y, t[4 * x + 2] = ta.z, t[4 * x + 3] = ta.w;
        }
        v = fast_log2_dev(dot20_scalar_dev(q, t), a.lg2, a.diff);
      }
      S[s] = v;
      if (a.Sss) {
        // S_ss[step] = ScoreSS(q, t, i, j) for the match steps, 0 elsewhere (:225-235); the column index sits in the meta word of
        // the record the profile came from; summed in step order by hhv_scorr_kernel
        float vs = 0.0f;
        if (st == 2 && i >= 1 && j >= 1) {
          const int32_t meta = __builtin_bit_cast(int32_t, a.records[(size_t)(rec0 + j) * REC_DW + REC_META]);
          vs = a.ss_table[a.ss_q_off[i - 1] + ((meta >> a.ss_t_shift) & a.ss_t_mask)];
        }
        a.Sss[po + s] = vs;
      }
    }
  }
}

// Kernel 2c: the second half of ScoreForBacktrace - one LANE per template: score = Viterbi score [- score_ss] + corr * Scorr
// with Scorr = sum over d = 1..4 of sum over steps of S[step] * S[step - d], ONE float accumulator through the four loops
// in the reference's order (:241-249) - 4 x nsteps dependent additions per template, which one lane of a wave per template
// used to walk alone (half of the old rescoring kernel's time).  Here 64 templates share a wave.  A lane reads ITS OWN
// template's S row, 64 steps (256 contiguous bytes, sixteen 16-byte loads) at a time, into registers - the values land in the
// lane that sums them, nothing is transposed; the tile after the current one is in flight while the current one is summed.
// (A first version moved 64 x 64 tiles through LDS with one coalesced load per template: 64 scalar address computations per
// tile made it issue bound - 0.2 ms per 100 k templates; profiles/r4_ab.txt.)
constexpr int SCORR_TILE = 64;
struct ScorrState {
  float acc, p1, p2, p3, p4;  // the accumulator; S[s - 1] .. S[s - 4]
};
// one tile of the loop over steps for lag D: steps s0 .. s0 + 63, of which [lo, lim] (tile-relative) take part
template <int D>
__device__ __forceinline__ void scorr_tile(ScorrState& z, const float4 (&t)[SCORR_TILE / 4], int lo, int lim) {
#pragma unroll
  for (int u = 0; u < SCORR_TILE; ++u) {
    const float4 q = t[u >> 2];
    const float cur = (u & 3) == 0 ? q.x : (u & 3) == 1 ? q.y : (u & 3) == 2 ? q.z : q.w;
    const float prev = D == 1 ? z.p1 : D == 2 ? z.p2 : D == 3 ? z.p3 : z.p4;
    const float term = cur * prev;
    z.acc = (u >= lo && u <= lim) ? z.acc + term : z.acc;
    z.p4 = z.p3, z.p3 = z.p2, z.p2 = z.p1, z.p1 = cur;
  }
}
__global__ void __launch_bounds__(64) hhv_scorr_kernel(TraceArgs a) {
  const int lane = threadIdx.x;
  const int k = blockIdx.x * LANES + lane;
  const bool valid = k < a.n;
  const int ns = valid ? a.hits[k].nsteps : 0;
  const int64_t po = valid ? a.path_off[k] : 0;
  int max_ns = ns;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) max_ns = max(max_ns, __shfl_xor(max_ns, o));
  const int n_tiles = max_ns / SCORR_TILE + 1;  // tile x = steps 64 x .. 64 x + 63 (step 0 is the unused entry of the pool)
  // a lane never reads beyond the tile that holds its own last step (the pool is followed by the next template's, the last
  // pool by 64 entries of slack: ensure_paths); what lies behind its last step is masked out of the sums
  const float4* const row = reinterpret_cast<const float4*>(a.S + po);  // (pools start on multiples of four entries)
  // secondary structure: one more pass over the tiles, of the S_ss row hhv_rescore_kernel wrote - score_ss = the sum of its
  // entries in step order (:225-235: the reference adds the match steps' values as it meets them; the zeros in between change nothing)
  const float4* const row_ss = a.Sss ? reinterpret_cast<const float4*>(a.Sss + po) : row;
  const int last_tile = ns / SCORR_TILE;
  float4 v[2][SCORR_TILE / 4];
  auto fetch = [&](const int seq, float4 (&dst)[SCORR_TILE / 4]) __attribute__((always_inline)) {
    const float4* p = (seq >= 4 * n_tiles ? row_ss : row) + (size_t)min(seq % n_tiles, last_tile) * (SCORR_TILE / 4);
#pragma unroll
    for (int x = 0; x < SCORR_TILE / 4; ++x) dst[x] = p[x];
  };
  ScorrState z = {0.f, 0.f, 0.f, 0.f, 0.f};
  float score_ss = 0.0f;
  const int total = (a.Sss ? 5 : 4) * n_tiles;  // tile sequence: d = 1 tiles 0 .. n_tiles-1, d = 2 the same tiles again, ... [, the S_ss tiles]
  auto sum_tile = [&](const int seq, const float4 (&t)[SCORR_TILE / 4]) __attribute__((always_inline)) {
    const int d = 1 + seq / n_tiles, tile_no = seq % n_tiles;
    const int lo = tile_no == 0 ? d + 1 : 0;             // the loop of lag d starts at step d + 1 (:241-249)
    const int lim = ns - tile_no * SCORR_TILE;           // last step of this lane, tile-relative
    switch (d) {
      case 1: scorr_tile<1>(z, t, lo, lim); break;
      case 2: scorr_tile<2>(z, t, lo, lim); break;
      case 3: scorr_tile<3>(z, t, lo, lim); break;
      case 4: scorr_tile<4>(z, t, lo, lim); break;
      default: {  // S_ss: steps 1 .. ns
        const int lo1 = tile_no == 0 ? 1 : 0;
#pragma unroll
        for (int u = 0; u < SCORR_TILE; ++u) {
          const float4 q = t[u >> 2];
          const float cur = (u & 3) == 0 ? q.x : (u & 3) == 1 ? q.y : (u & 3) == 2 ? q.z : q.w;
          score_ss = (u >= lo1 && u <= lim) ? score_ss + cur : score_ss;
        }
      }
    }
  };
  fetch(0, v[0]);
  for (int seq = 0; seq < total; seq += 2) {
    if (seq + 1 < total) fetch(seq + 1, v[1]);
    sum_tile(seq, v[0]);
    if (seq + 1 >= total) break;
    if (seq + 2 < total) fetch(seq + 2, v[0]);
    sum_tile(seq + 1, v[1]);
  }
  const float Scorr = z.acc;
  if (!valid) return;
  float score = a.hits[k].viterbi_score;
  // :225-238: score_ss = sum over MM steps of ScoreSS(q,t,i,j) in step order (summed above); subtracted when ssm == 2
  if (a.ss_mode == 2) score -= score_ss;
  score += a.corr * Scorr;
  a.hits[k].score_ss = score_ss;
  a.hits[k].score = score;
}

// ---------------------------------------------------------------------------------------------
// host-side launch helpers

void* stream_kernel_w64(int R, bool local, bool bt, bool celloff, bool multi, bool ss, bool first_strip) {
  return stream_kernel_pick<LANES>(R, local, bt, celloff, multi, ss, first_strip);
}

static void* pick(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss, bool first_strip = false) {
  if (W == LANES) return stream_kernel_w64(R, local, bt, celloff, multi, ss, first_strip);
  if (multi) return nullptr;
  if (W == 32) return stream_kernel_w32(R, local, bt, celloff, ss);
  if (W == 16) return stream_kernel_w16(R, local, bt, celloff, ss);
  return nullptr;
}

// ev_start / ev_stop (hipEvent_t or null): attached to the kernel's own dispatch (hipExtLaunchKernel) - its begin and end time stamps,
// without the barrier packets of hipEventRecord in front of and behind the launch (5 - 6 us each between two kernels of a stream)
int launch_stream(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss, const StreamArgs& a, int n_waves,
                  void* stream, void* ev_start, void* ev_stop) {
  // (the first strip of a multi-strip plan has a kernel of its own: no code of the later strips in its steps)
  void* fn = pick(W, R, local, bt, celloff, multi, ss, multi && a.pass_first != 0);
  if (!fn) return -1;
  StreamArgs args = a;
  void* kargs[] = {&args};
  // (the 64-lane secondary-structure variants are workgroups of SS_WAVES wavefronts sharing one LDS table: hhv_ss_kernel)
  const bool wide = ss && W == LANES;
  const dim3 grid(wide ? (n_waves + SS_WAVES - 1) / SS_WAVES : n_waves), block(wide ? SS_WAVES * LANES : LANES);
  hipError_t e = (ev_start || ev_stop) ? hipExtLaunchKernel(fn, grid, block, kargs, 0, (hipStream_t)stream, (hipEvent_t)ev_start, (hipEvent_t)ev_stop, 0)
                                       : hipLaunchKernel(fn, grid, block, kargs, 0, (hipStream_t)stream);
  return e == hipSuccess ? 0 : -(int)e;
}

// wavefronts per workgroup of the kernel launch_stream starts: the caller rounds its wave count up to a multiple
int stream_kernel_waves(int W, bool ss) { return ss && W == LANES ? SS_WAVES : 1; }

int stream_kernel_occupancy(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss, int* blocks_per_cu,
                            int* vgprs) {
  void* fn = pick(W, R, local, bt, celloff, multi, ss);
  if (!fn) return -1;
  int nb = 0;
  const int wpw = stream_kernel_waves(W, ss);
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, wpw * LANES, 0);
  if (e != hipSuccess) return -(int)e;
  hipFuncAttributes fa;
  e = hipFuncGetAttributes(&fa, fn);
  if (e != hipSuccess) return -(int)e;
  if (blocks_per_cu) *blocks_per_cu = nb * wpw;  // resident WAVEFRONTS (= one-wave workgroups for every kernel but hhv_ss_kernel)
  if (vgprs) *vgprs = fa.numRegs;
  return 0;
}

// ---- the paths of a set, packed for the host (hhv_hit_paths_packed) -----------------------------------------------------------
// The trace pool holds Lq + Lt + 2 entries per template (its capacity), 13 bytes each in four arrays: 156 MB for 20 000 templates of
// 300 columns, of which a third is path.  One wavefront per hit copies entries 0 .. nsteps of its path into ONE compact record
// stream - i and j as 16-bit values (template columns end at 65 535, query rows at 32 767), the state byte, the score - at the
// offset the host computed from the hits' step counts: 9 bytes per step to move instead of 13 per pool entry.
__global__ void __launch_bounds__(64) hhv_pack_paths_kernel(const DevHit* __restrict__ hits, int n, const int64_t* __restrict__ path_off,
                                                            const int32_t* __restrict__ pi, const int32_t* __restrict__ pj,
                                                            const int8_t* __restrict__ ps, const float* __restrict__ pS,
                                                            const int64_t* __restrict__ out_off, uint16_t* __restrict__ oi,
                                                            uint16_t* __restrict__ oj, int8_t* __restrict__ os, float* __restrict__ oS) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int ns = hits[k].nsteps;
  const int64_t src = path_off[k], dst = out_off[k];
  for (int e = threadIdx.x; e <= ns; e += 64) {
    oi[dst + e] = e ? (uint16_t)pi[src + e] : (uint16_t)0;
    oj[dst + e] = e ? (uint16_t)pj[src + e] : (uint16_t)0;
    os[dst + e] = e ? ps[src + e] : (int8_t)0;
    oS[dst + e] = e ? pS[src + e] : 0.0f;
  }
}
int launch_pack_paths(const DevHit* hits, int n, const int64_t* path_off, const int32_t* pi, const int32_t* pj, const int8_t* ps,
                      const float* pS, const int64_t* out_off, uint16_t* oi, uint16_t* oj, int8_t* os, float* oS, void* stream) {
  hipLaunchKernelGGL(hhv_pack_paths_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, hits, n, path_off, pi, pj, ps, pS, out_off, oi, oj, os, oS);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int launch_trace(const TraceArgs& a, void* stream) {
  // The walk: one wavefront per template (a round trip per run of the path; hhv_trace_wave_kernel) - measured ahead of one lane
  // per template at every set size (10 000 templates: the whole backtrace step 2.40 -> 2.27 ms, 100 000: 19.26 -> 19.01;
  // profiles/r5_ab.txt ab-r5-2).  a.trace_mode 0 (hhv_set_launch_policy) selects the lane-per-template kernel: tests, measurements.
  const bool per_wave = a.trace_mode != 0;
  {
    const dim3 grid(per_wave ? a.n : (a.n + LANES - 1) / LANES), block(LANES);
    hipStream_t st = (hipStream_t)stream;
    const int R = a.plan.P == 1 ? a.plan.R_hi : 0;
#define HHV_TRACE_CASE(r, m)                                                       \
  if (R == r && a.bt_mm == m) {                                                    \
    if (per_wave) hipLaunchKernelGGL((hhv_trace_wave_kernel<r, m>), grid, block, 0, st, a); \
    else hipLaunchKernelGGL((hhv_trace_kernel<r, m>), grid, block, 0, st, a);      \
  } else
#define HHV_TRACE_ROWS(r) HHV_TRACE_CASE(r, BT_MM_RUNNING) HHV_TRACE_CASE(r, BT_MM_FIRST_EQUAL) HHV_TRACE_CASE(r, BT_MM_FIRST_EQUAL_NEG)
    HHV_TRACE_ROWS(1) HHV_TRACE_ROWS(2) HHV_TRACE_ROWS(3) HHV_TRACE_ROWS(4) HHV_TRACE_ROWS(5) {
      if (per_wave) hipLaunchKernelGGL((hhv_trace_wave_kernel<0, 0>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((hhv_trace_kernel<0, 0>), grid, block, 0, st, a);
    }
#undef HHV_TRACE_ROWS
#undef HHV_TRACE_CASE
  }
  if (!per_wave) hipLaunchKernelGGL(hhv_rescore_kernel, dim3(a.n), dim3(LANES), 0, (hipStream_t)stream, a);  // (the wavefront walk writes (i, j) and S itself)
  hipLaunchKernelGGL(hhv_scorr_kernel, dim3((a.n + LANES - 1) / LANES), dim3(LANES), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv
