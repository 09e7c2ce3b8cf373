// hhv_topk.hip -- device-side selection of the K best hits (the per-GPU half of the sharded top-K
// merge).  Key = Hit.score descending, ties broken by the smaller template index, the
// order ViterbiRunner's caller establishes when it sorts the hit list (src/hhhit.h:116-126 compares
// score_aass = -score).  A full 64-bit radix sort of (orderable score, ~index) via hipCUB is used:
// n <= a few 10^5 per GPU, this stage is microseconds next to the DP.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <string>

#include "hhv_internal.h"
#include "viterbi_lane.h"

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

// ---- first selection step of the prefilter on the device (Prefilter::prefilter_db, src/hhprefilter.cpp:461-505) --------
namespace hhv {

// util-inl.h:83-93: the polynomial constants are double literals there - double Horner chain, rounded to float once
__device__ __forceinline__ float flog2_dev(float x) {
  if (x <= 0) return -128;
  uint32_t bits = __float_as_uint(x);
  const float e = (float)((int)((bits & 0x7F800000u) >> 23) - 0x7f);
  x = __uint_as_float((bits & 0x007FFFFFu) | 0x3f800000u);
  x = (float)((double)x - 1.0);
  x = (float)((double)x * (1.441740 + (double)x * (-0.7077702 + (double)x * (0.4123442 + (double)x * (-0.1903190 + (double)x * 0.0440047)))));
  return x + e;
}

// key = (length-corrected score, id), both descending like the reference's sort + reverse of (score, id) pairs;
// above[0] counts the sequences above the threshold
__global__ void pf_select_keys_kernel(const int32_t* __restrict__ scores, const int64_t* __restrict__ offsets, int n,
                                      float log_qlen, int bit_factor, int smax_thresh, uint64_t* __restrict__ keys,
                                      unsigned int* __restrict__ above) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  bool hit = false;
  if (k < n) {
    const int len = (int)(offsets[k + 1] - offsets[k]);
    const int corrected = scores[k] - (int)((float)bit_factor * (log_qlen + flog2_dev((float)len)));
    keys[k] = ((uint64_t)((uint32_t)corrected ^ 0x80000000u) << 32) | (uint32_t)k;
    hit = corrected > smax_thresh;
  }
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(above, (unsigned int)__popcll(m));
}

__global__ void pf_select_ids_kernel(const uint64_t* __restrict__ sorted, int m, int32_t* __restrict__ ids) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < m) ids[k] = (int32_t)(sorted[k] & 0xFFFFFFFFu);
}

// keys/sorted: n uint64, temp: topk_temp_bytes(n), above: one zeroed uint.  Asynchronous on `stream`.
int pf_select_sort(const int32_t* d_scores, const int64_t* d_offsets, int n, float log_qlen, int bit_factor, int smax_thresh,
                   uint64_t* keys, uint64_t* sorted, void* temp, size_t temp_bytes, unsigned int* above, hipStream_t stream) {
  const int threads = 256;
  hipLaunchKernelGGL(pf_select_keys_kernel, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, d_scores, d_offsets, n,
                     log_qlen, bit_factor, smax_thresh, keys, above);
  const hipError_t e = hipcub::DeviceRadixSort::SortKeysDescending(temp, temp_bytes, keys, sorted, n, 0, 64, stream);
  return e == hipSuccess ? 0 : -(int)e;
}

int pf_select_ids(const uint64_t* sorted, int m, int32_t* d_ids, hipStream_t stream) {
  const int threads = 256;
  hipLaunchKernelGGL(pf_select_ids_kernel, dim3((m + threads - 1) / threads), dim3(threads), 0, stream, sorted, m, d_ids);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv

// ---- device-side subset of a resident template set: copy the records of the selected templates, renumber headers -------
namespace hhv {

__global__ void __launch_bounds__(256) tset_gather_kernel(const float* __restrict__ src, const int64_t* __restrict__ src_off,
                                                          const int32_t* __restrict__ ids, const int64_t* __restrict__ dst_off,
                                                          const int32_t* __restrict__ L, float* __restrict__ dst) {
  const int k = blockIdx.x;  // slot in the new set
  const float4* s = reinterpret_cast<const float4*>(src + (size_t)src_off[ids[k]] * REC_DW);
  float4* d = reinterpret_cast<float4*>(dst + (size_t)dst_off[k] * REC_DW);
  const int n4 = (L[k] + 1) * (REC_DW / 4);
  for (int e = threadIdx.x; e < n4; e += 256) {
    float4 v = s[e];
    if (e == 0) v.x = __int_as_float(k);  // header record: [0] = template index inside the set
    d[e] = v;
  }
}

int tset_gather(const float* src, const int64_t* src_off, const int32_t* ids, const int64_t* dst_off, const int32_t* L, int n,
                float* dst, hipStream_t stream) {
  hipLaunchKernelGGL(tset_gather_kernel, dim3(n), dim3(256), 0, stream, src, src_off, ids, dst_off, L, dst);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv
