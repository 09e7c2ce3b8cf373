// hhv_topk.hip -- device-side selection of the K best hits (the per-GPU half of the sharded top-K
// merge).  Key = Hit.score descending, ties broken by the smaller template index, the
// order ViterbiRunner's caller establishes when it sorts the hit list (src/hhhit.h:116-126 compares
// score_aass = -score).  A full 64-bit radix sort of (orderable score, ~index) via hipCUB is used:
// n <= a few 10^5 per GPU, this stage is microseconds next to the DP.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <string>

#include "hhv_internal.h"

namespace hhv {

__global__ void topk_keys_kernel(const DevHit* __restrict__ hits, int n, uint64_t* __restrict__ keys) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t u = __builtin_bit_cast(uint32_t, hits[k].score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
  keys[k] = ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)k);
}

__global__ void topk_gather_kernel(const DevHit* __restrict__ hits, const uint64_t* __restrict__ sorted, int k,
                                   DevHit* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(sorted[t] & 0xFFFFFFFFu);
  out[t] = hits[idx];
}

__global__ void results_to_hits_kernel(const DevResult* __restrict__ res, int n, DevHit* __restrict__ hits) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const DevResult r = res[k];
  DevHit h;
  h.score = r.score;
  h.viterbi_score = r.score;
  h.score_ss = 0.0f;
  h.index = k;
  h.i1 = h.j1 = 0;
  h.i2 = r.i2;
  h.j2 = r.j2;
  h.nsteps = 0;
  h.matched_cols = 0;
  hits[k] = h;
}

size_t topk_temp_bytes(int n) {
  size_t bytes = 0;
  (void)hipcub::DeviceRadixSort::SortKeysDescending(nullptr, bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, n, 0, 64,
                                                    (hipStream_t)0);
  return bytes ? bytes : 1;
}

void results_to_hits(const DevResult* d_res, int n, DevHit* d_hits, hipStream_t stream) {
  const int threads = 256;
  hipLaunchKernelGGL(results_to_hits_kernel, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, d_res, n,
                     d_hits);
}

// keys/sorted: n uint64 each, temp: topk_temp_bytes(n).  Asynchronous on `stream`.
int topk_device(const DevHit* d_hits, int n, int k, DevHit* d_out, uint64_t* keys, uint64_t* sorted, void* temp,
                size_t temp_bytes, hipStream_t stream, std::string* err) {
  const int threads = 256;
  hipLaunchKernelGGL(topk_keys_kernel, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, d_hits, n, keys);
  hipError_t e = hipcub::DeviceRadixSort::SortKeysDescending(temp, temp_bytes, keys, sorted, n, 0, 64, stream);
  if (e != hipSuccess) {
    if (err) *err = std::string("radix sort: ") + hipGetErrorString(e);
    return -1;
  }
  hipLaunchKernelGGL(topk_gather_kernel, dim3((k + threads - 1) / threads), dim3(threads), 0, stream, d_hits, sorted, k,
                     d_out);
  e = hipGetLastError();
  if (e != hipSuccess) {
    if (err) *err = std::string("gather: ") + hipGetErrorString(e);
    return -1;
  }
  return 0;
}

}  // namespace hhv
