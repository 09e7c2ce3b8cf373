// hhv_prefilter.hip -- HHblits prefilter kernels on MI355X: replaces
//   Prefilter::ungapped_sse_score   src/hhprefilter.cpp:214-278   gapless profile-vs-sequence score
//   Prefilter::swStripedByte        src/hhprefilter.cpp:70-212    striped Smith-Waterman (Farrar / Zhao)
// for one query column-state profile against a resident database of column-state sequences.
//
// The reference vectorises along the QUERY with 32 unsigned bytes per AVX2 register ("striped": vector
// element k of segment row j is query position k*W + j, W = ceil(Lq/32)).  Here a 32-lane half-wavefront IS
// that vector: lane k owns the k-th stripe element, one cell per lane per inner iteration, so the recurrence -
// including the properties that depend on the striping (E is updated before the lazy-F correction, the F
// chain is restarted per segment in the main loop) - is reproduced by construction, with 32-bit lanes doing
// the saturating uint8 arithmetic.  The query profile sits striped in LDS (one broadcast-free byte per lane),
// the H/E columns of each sequence in LDS too; a 256-thread block runs 8 sequences at a time.
#include <hip/hip_runtime.h>

#include "hhv_internal.h"

namespace hhv {

__device__ __forceinline__ int shift_in_half(int v, int k) {
  // simdi8_shiftl(x, 1): element k <- element k-1, element 0 <- 0 (within the 32-lane half)
  const int up = __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
  return k == 0 ? 0 : up;
}

template <bool GAPPED>
__global__ void __launch_bounds__(256) hhv_prefilter_kernel(PrefilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = a.W;
  const int lane = threadIdx.x & 63;
  const int k = threadIdx.x & 31;        // stripe element
  const int half = threadIdx.x >> 5;     // 0..7: sequence slot inside the block
  unsigned char* sprof = smem;                                   // [220][W][32]
  unsigned char* state = smem + (size_t)220 * W * 32 + (size_t)half * 3 * W * 32;
  // stripe the plain [220][Lq] profile exactly like Prefilter::stripe_query_profile (:386-425)
  for (int e = threadIdx.x; e < 220 * W * 32; e += 256) {
    const int kk = e & 31, j = (e >> 5) % W, x = (e >> 5) / W;
    const int p = kk * W + j;
    sprof[e] = (p >= a.Lq) ? (unsigned char)a.offset : a.profile[(size_t)x * a.Lq + p];
  }
  __syncthreads();
  const int go = a.gap_init, ge = a.gap_extend, bias = a.offset;
  (void)lane;

  for (int64_t slot = (int64_t)blockIdx.x * 8 + half; slot < a.n_jobs; slot += (int64_t)gridDim.x * 8) {
    const int sid = a.subset ? a.subset[slot] : (int)slot;
    const unsigned char* seq = a.seqs + a.offsets[sid];
    const int len = (int)(a.offsets[sid + 1] - a.offsets[sid]);
    unsigned char* Ha = state;               // pvHStore
    unsigned char* Hb = state + W * 32;      // pvHLoad
    unsigned char* E = state + 2 * W * 32;
    for (int j = 0; j < W; ++j) {
      Ha[j * 32 + k] = 0;
      Hb[j * 32 + k] = 0;
      E[j * 32 + k] = 0;
    }
    int vmax = 0;
    for (int i = 0; i < len; ++i) {
      const unsigned char* P = sprof + (size_t)seq[i] * W * 32;
      if (GAPPED) {
        int F = 0;
        int H = shift_in_half(Ha[(W - 1) * 32 + k], k);
        unsigned char* t = Hb;  // swap the two H buffers (:129-132)
        Hb = Ha;
        Ha = t;
        for (int j = 0; j < W; ++j) {
          H = min(255, H + (int)P[j * 32 + k]);
          H = max(0, H - bias);
          int e = E[j * 32 + k];
          H = max(H, e);
          H = max(H, F);
          vmax = max(vmax, H);
          Ha[j * 32 + k] = (unsigned char)H;
          H = max(0, H - go);
          e = max(max(0, e - ge), H);
          E[j * 32 + k] = (unsigned char)e;
          F = max(max(0, F - ge), H);
          H = Hb[j * 32 + k];
        }
        // lazy-F loop (:176-203)
        int j = 0;
        H = Ha[k];
        F = shift_in_half(F, k);
        bool need = max(0, F - max(0, H - go)) != 0;
        while (((__ballot(need) >> (threadIdx.x & 32)) & 0xFFFFFFFFull) != 0) {
          H = max(H, F);
          vmax = max(vmax, H);
          Ha[j * 32 + k] = (unsigned char)H;
          F = max(0, F - ge);
          ++j;
          if (j >= W) {
            j = 0;
            F = shift_in_half(F, k);
          }
          H = Ha[j * 32 + k];
          need = max(0, F - max(0, H - go)) != 0;
        }
      } else {
        // ungapped (:246-274): s_curr = Ha, s_prev = Hb
        int S = shift_in_half(Ha[(W - 1) * 32 + k], k);
        unsigned char* t = Hb;
        Hb = Ha;
        Ha = t;
        for (int j = 0; j < W; ++j) {
          S = min(255, S + (int)P[j * 32 + k]);
          S = max(0, S - bias);
          Ha[j * 32 + k] = (unsigned char)S;
          vmax = max(vmax, S);
          S = Hb[j * 32 + k];
        }
      }
    }
    // horizontal maximum over the 32 stripe elements (simd_hmax)
    for (int o = 16; o >= 1; o >>= 1) vmax = max(vmax, __shfl_xor(vmax, o, 32));
    if (k == 0) a.scores[slot] = vmax;
  }
}

int launch_prefilter(const PrefilterArgs& a, bool gapped, int n_blocks, size_t lds_bytes, void* stream) {
  if (gapped) {
    (void)hipFuncSetAttribute((const void*)hhv_prefilter_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(hhv_prefilter_kernel<true>, dim3(n_blocks), dim3(256), lds_bytes, (hipStream_t)stream, a);
  } else {
    (void)hipFuncSetAttribute((const void*)hhv_prefilter_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(hhv_prefilter_kernel<false>, dim3(n_blocks), dim3(256), lds_bytes, (hipStream_t)stream, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv
