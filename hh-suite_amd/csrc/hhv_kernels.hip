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
// Kernel 2b hhv_rescore_kernel Viterbi::ScoreForBacktrace (src/hhviterbi.cpp:195-281), one wave per template.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (fp contraction would change results).
#include <hip/hip_runtime.h>

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
__device__ __forceinline__ void trace_step(int& state, int& i, int& j, int& matched, uint32_t b) {
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

// Kernel 2a: Viterbi::Backtrace (src/hhviterbi.cpp:83-160) - a serial pointer chase, one lane per template.
// Every step reads one 8-byte entry that depends on the step before - an HBM round trip per step if done naively
// (the entries of a launch are written once, gigabytes ago).  Single-pass plans therefore read a WINDOW per round trip:
// from cell (i, j) with entry (row, g) a path can only move to rows row, row - 1, row - 2 and to lane g or g - 1
// (bt_entry: row = record + lane; a step lowers i and / or j by one), so the ten entries {row .. row - 4} x {g - 1, g} -
// five independent 16-byte reads - cover the next two steps at least, four on a diagonal; the walk continues out of
// registers until it leaves the window.
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
  if (a.plan.P == 1) {
    const int R = a.plan.R_hi, W = a.plan.W;
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
          b = bt_decode(entry, rr, R, a.bt_mm);
        }
        step++;
        states[step] = (int8_t)state;
        i_steps[step] = i;
        j_steps[step] = j;
        trace_step(state, i, j, matched, b);
      }
    }
  } else {
    while (state != 0) {
      step++;
      states[step] = (int8_t)state;
      i_steps[step] = i;
      j_steps[step] = j;
      uint32_t b = 0;
      if (i >= 1 && j >= 1) {
        int pass, g, rr, Rp;
        a.plan.locate(i, pass, g, rr, Rp);
        b = bt_decode(a.bt[(size_t)pass * a.bt_pass_stride + bt_entry(rec0 + j, g, a.plan.W)], rr, Rp, a.bt_mm);
      }
      trace_step(state, i, j, matched, b);
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

void* stream_kernel_w64(int R, bool local, bool bt, bool celloff, bool multi, bool ss) {
  return stream_kernel_pick<LANES>(R, local, bt, celloff, multi, ss);
}

static void* pick(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss) {
  if (W == LANES) return stream_kernel_w64(R, local, bt, celloff, multi, ss);
  if (multi) return nullptr;
  if (W == 32) return stream_kernel_w32(R, local, bt, celloff, ss);
  if (W == 16) return stream_kernel_w16(R, local, bt, celloff, ss);
  return nullptr;
}

int launch_stream(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss, const StreamArgs& a, int n_waves,
                  void* stream) {
  void* fn = pick(W, R, local, bt, celloff, multi, ss);
  if (!fn) return -1;
  StreamArgs args = a;
  void* kargs[] = {&args};
  hipError_t e = hipLaunchKernel(fn, dim3(n_waves), dim3(LANES), kargs, 0, (hipStream_t)stream);
  return e == hipSuccess ? 0 : -(int)e;
}

int stream_kernel_occupancy(int W, int R, bool local, bool bt, bool celloff, bool multi, bool ss, int* blocks_per_cu,
                            int* vgprs) {
  void* fn = pick(W, R, local, bt, celloff, multi, ss);
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
