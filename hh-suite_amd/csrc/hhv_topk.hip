// hhv_topk.hip -- device-side selection of the K best hits (the per-GPU half of the sharded top-K
// merge).  Key = Hit.score descending, ties broken by the smaller template index, the
// order ViterbiRunner's caller establishes when it sorts the hit list (src/hhhit.h:116-126 compares
// score_aass = -score).  A full 64-bit radix sort of (orderable score, ~index) via hipCUB is used:
// n <= a few 10^5 per GPU, this stage is microseconds next to the DP.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
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

// gids: the shard's global template ids (hhv_tset_set_global_ids), or null = the index inside the set
__global__ void topk_gather_kernel(const DevHit* __restrict__ hits, const uint64_t* __restrict__ sorted, int k,
                                   const int32_t* __restrict__ gids, DevHit* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(sorted[t] & 0xFFFFFFFFu);
  DevHit h = hits[idx];
  if (gids) h.index = gids[h.index];
  out[t] = h;
}

// ---- merge of the hit lists of several shards (the other half of SURVEY.md 8e) ---------------------------------------
// in: m records, the concatenation of every shard's hhv_topk output (global ids in `index`, padding records index < 0).
// out: the k best, score descending, ties by the smaller global id - the order the reference's caller gives the hit list
// (src/hhhit.h:116-126) after ViterbiRunner::alignment appended the batches serially (src/hhviterbirunner.cpp:173).
// One workgroup: m <= MERGE_MAX keys are sorted in LDS by a bitonic network (m = ranks x K, a few thousand).
constexpr int MERGE_MAX = 4096;

__device__ __forceinline__ uint64_t merge_key(const DevHit& h) {
  if (h.index < 0) return 0;  // padding sorts last
  uint32_t u = __builtin_bit_cast(uint32_t, h.score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)h.index);  // valid records: low word >= 0x80000000
}

__global__ void __launch_bounds__(1024) merge_hits_kernel(const DevHit* __restrict__ in, int m, int k, DevHit* __restrict__ out,
                                                          int* __restrict__ n_out) {
  __shared__ uint64_t key[MERGE_MAX];
  __shared__ uint16_t pos[MERGE_MAX];
  __shared__ int n_valid;
  int P = 2;
  while (P < m) P <<= 1;
  if (threadIdx.x == 0) n_valid = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    const uint64_t kk = i < m ? merge_key(in[i]) : 0;
    key[i] = kk;
    pos[i] = (uint16_t)i;
    mine += kk != 0;
  }
  if (mine) atomicAdd(&n_valid, mine);
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < P / 2; i += blockDim.x) {
        const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
        const bool desc = (lo & size) == 0;  // final order: descending
        const uint64_t a = key[lo], b = key[hi];
        if ((a < b) == desc) {
          key[lo] = b;
          key[hi] = a;
          const uint16_t t = pos[lo];
          pos[lo] = pos[hi];
          pos[hi] = t;
        }
      }
    }
  }
  __syncthreads();
  const int nv = min(n_valid, k);
  for (int t = threadIdx.x; t < k; t += blockDim.x) {
    DevHit h;
    if (t < nv) {
      h = in[pos[t]];
    } else {
      h.score = h.viterbi_score = h.score_ss = __builtin_bit_cast(float, 0xFFFFFFFFu);
      h.index = h.i1 = h.j1 = h.i2 = h.j2 = h.nsteps = h.matched_cols = -1;
    }
    out[t] = h;
  }
  if (threadIdx.x == 0) *n_out = nv;
}

// general case (m > MERGE_MAX): keys to global memory, hipCUB pair sort
__global__ void merge_keys_kernel(const DevHit* __restrict__ in, int m, uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                  int* __restrict__ n_valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  if (i < m) {
    const uint64_t kk = merge_key(in[i]);
    keys[i] = kk;
    vals[i] = (uint32_t)i;
    ok = kk != 0;
  }
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_valid, (int)__popcll(b));
}
__global__ void merge_gather_kernel(const DevHit* __restrict__ in, const uint32_t* __restrict__ order, int k, int* __restrict__ n_valid,
                                    DevHit* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k) return;
  const int nv = min(*n_valid, k);
  DevHit h;
  if (t < nv) {
    h = in[order[t]];
  } else {
    h.score = h.viterbi_score = h.score_ss = __builtin_bit_cast(float, 0xFFFFFFFFu);
    h.index = h.i1 = h.j1 = h.i2 = h.j2 = h.nsteps = h.matched_cols = -1;
  }
  out[t] = h;
}

__global__ void merge_fix_count_kernel(int* n_valid, int k) { *n_valid = min(*n_valid, k); }

// d_n: one device int (receives min(k, valid records)).  Asynchronous on `stream`.
int merge_hits_device(const DevHit* d_in, int m, int k, DevHit* d_out, int* d_n, hipStream_t stream, std::string* err) {
  if (m <= MERGE_MAX) {
    hipLaunchKernelGGL(merge_hits_kernel, dim3(1), dim3(1024), 0, stream, d_in, m, k, d_out, d_n);
  } else {
    uint64_t *keys = nullptr, *keys2 = nullptr;
    uint32_t *vals = nullptr, *vals2 = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairsDescending(nullptr, temp_bytes, keys, keys2, vals, vals2, m, 0, 64, stream);
    hipError_t e = hipMalloc(&keys, (size_t)m * 8);
    if (e == hipSuccess) e = hipMalloc(&keys2, (size_t)m * 8);
    if (e == hipSuccess) e = hipMalloc(&vals, (size_t)m * 4);
    if (e == hipSuccess) e = hipMalloc(&vals2, (size_t)m * 4);
    if (e == hipSuccess) e = hipMalloc(&temp, temp_bytes ? temp_bytes : 1);
    if (e == hipSuccess) e = hipMemsetAsync(d_n, 0, sizeof(int), stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(merge_keys_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, d_in, m, keys, vals, d_n);
      e = hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, keys, keys2, vals, vals2, m, 0, 64, stream);
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(merge_gather_kernel, dim3((k + 255) / 256), dim3(256), 0, stream, d_in, vals2, k, d_n, d_out);
      hipLaunchKernelGGL(merge_fix_count_kernel, dim3(1), dim3(1), 0, stream, d_n, k);
      e = hipStreamSynchronize(stream);
    }
    (void)hipFree(keys);
    (void)hipFree(keys2);
    (void)hipFree(vals);
    (void)hipFree(vals2);
    (void)hipFree(temp);
    if (e != hipSuccess) {
      if (err) *err = std::string("merge (general path): ") + hipGetErrorString(e);
      return -1;
    }
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (err) *err = std::string("merge: ") + hipGetErrorString(e);
    return -1;
  }
  return 0;
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
int topk_device(const DevHit* d_hits, int n, int k, const int32_t* gids, DevHit* d_out, uint64_t* keys, uint64_t* sorted,
                void* temp, size_t temp_bytes, hipStream_t stream, std::string* err) {
  const int threads = 256;
  hipLaunchKernelGGL(topk_keys_kernel, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, d_hits, n, keys);
  hipError_t e = hipcub::DeviceRadixSort::SortKeysDescending(temp, temp_bytes, keys, sorted, n, 0, 64, stream);
  if (e != hipSuccess) {
    if (err) *err = std::string("radix sort: ") + hipGetErrorString(e);
    return -1;
  }
  hipLaunchKernelGGL(topk_gather_kernel, dim3((k + threads - 1) / threads), dim3(threads), 0, stream, d_hits, sorted, k,
                     gids, d_out);
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

// ---- global mode: which templates are shorter than the longest of their SIMD batch in the reference ----------------------
namespace hhv {
__global__ void __launch_bounds__(256) header_flag_kernel(float* __restrict__ records, const int64_t* __restrict__ rec_off,
                                                          const unsigned char* __restrict__ flags, int n) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= n) return;
  int32_t* meta = reinterpret_cast<int32_t*>(records + (size_t)rec_off[k] * REC_DW + REC_META);
  *meta = META_HDR | ((flags && flags[k]) ? META_NOLASTCOL : 0);
}
int set_header_flags(float* records, const int64_t* rec_off, const unsigned char* flags, int n, hipStream_t stream) {
  hipLaunchKernelGGL(header_flag_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, records, rec_off, flags, n);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}
}  // namespace hhv

// ---- cell-off masks of the alternative-alignment rounds, built on the device ------------------------------------------
// Viterbi::ExcludeAlignment (src/hhviterbi.cpp:61-77): for every step but the last of an earlier alignment, the cells
// (i +- 40, j) and (i, j +- 40) are switched off.  The mask is an input of the stream kernel: bit 7 of byte r of the
// backtrace entry [pass][record][lane] (viterbi_lane.h).
namespace hhv {

struct CellOffGeom {
  uint64_t* bt;
  const int64_t* rec_off;
  const int32_t* L;
  int64_t pass_stride;  // entries per pass
  int Lq;
  StripPlan plan;
  int bt_mm;  // bt_matrix_kernel only: encoding of the MM predecessor in the entries
};
__device__ __forceinline__ void celloff_set(const CellOffGeom& g, int t, int i, int j) {
  int pass, lane, r, Rp;
  g.plan.locate(i, pass, lane, r, Rp);
  atomicOr((unsigned long long*)(g.bt + (size_t)pass * g.pass_stride + bt_entry(g.rec_off[t] + j, lane, g.plan.W)),
           0x80ull << (8 * r));
}

// one workgroup per template: clear every entry, then the -excl / -template_excl ranges
__global__ void __launch_bounds__(256) celloff_clear_kernel(CellOffGeom g, const int32_t* __restrict__ ranges, int n_q, int n_t) {
  const int t = blockIdx.x, Lt = g.L[t], W = g.plan.W;
  for (int pass = 0; pass < g.plan.P; ++pass) {
    uint64_t* e = g.bt + (size_t)pass * g.pass_stride;
    for (int k = threadIdx.x; k < Lt * W; k += 256) e[bt_entry(g.rec_off[t] + 1 + k / W, k % W, W)] = 0;
  }
  __syncthreads();
  for (int q = 0; q < n_q; ++q) {
    const int lo = ranges[2 * q], hi = min(ranges[2 * q + 1], g.Lq);
    for (int c = threadIdx.x; c < (hi - lo + 1) * Lt; c += 256) celloff_set(g, t, lo + c / Lt, 1 + c % Lt);
  }
  for (int q = 0; q < n_t; ++q) {
    const int lo = ranges[2 * (n_q + q)], hi = min(ranges[2 * (n_q + q) + 1], Lt);
    for (int c = threadIdx.x; c < (hi - lo + 1) * g.Lq; c += 256) celloff_set(g, t, 1 + c % g.Lq, lo + c / g.Lq);
  }
}

// hhv_set_celloff: the caller's byte mask of ONE template -> its entries (every entry of the template is rewritten)
__global__ void __launch_bounds__(256) celloff_mask_kernel(CellOffGeom g, int t, int Lt, const unsigned char* __restrict__ mask) {
  const int W = g.plan.W;
  for (int pass = 0; pass < g.plan.P; ++pass) {
    const int R = g.plan.R(pass), ilo = g.plan.base(pass) + 1;
    uint64_t* e = g.bt + (size_t)pass * g.pass_stride;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < Lt * W; k += gridDim.x * 256) {
      const int j = 1 + k / W, lane = k % W;
      uint64_t v = 0;
      if (mask)
        for (int r = 0; r < R; ++r) {
          const int i = ilo + lane * R + r;
          if (i <= g.Lq && mask[(size_t)i * (Lt + 1) + j]) v |= 0x80ull << (8 * r);
        }
      e[bt_entry(g.rec_off[t] + j, lane, W)] = v;
    }
  }
}

// hhv_backtrace_matrix: the entries of one template decoded into the reference's byte matrix [(Lq+1)][(Lt+1)]
__global__ void __launch_bounds__(256) bt_matrix_kernel(CellOffGeom g, int t, int Lt, unsigned char* __restrict__ out) {
  const int64_t cells = (int64_t)(g.Lq + 1) * (Lt + 1);
  for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < cells; c += (int64_t)gridDim.x * 256) {
    const int i = (int)(c / (Lt + 1)), j = (int)(c % (Lt + 1));
    unsigned char b = 0;
    if (i >= 1 && j >= 1) {
      int pass, lane, r, Rp;
      g.plan.locate(i, pass, lane, r, Rp);
      b = (unsigned char)bt_decode(g.bt[(size_t)pass * g.pass_stride + bt_entry(g.rec_off[t] + j, lane, g.plan.W)], r, Rp, g.bt_mm);
    }
    out[c] = b;
  }
}

// one workgroup per earlier alignment: its +-40 cross
__global__ void __launch_bounds__(256) celloff_paths_kernel(CellOffGeom g, const int32_t* __restrict__ template_of,
                                                            const int64_t* __restrict__ path_off,
                                                            const int32_t* __restrict__ pi, const int32_t* __restrict__ pj) {
  const int p = blockIdx.x, t = template_of[p], Lt = g.L[t];
  const int64_t o = path_off[p];
  const int ns = (int)(path_off[p + 1] - o) - 1;  // the last step is skipped, like the reference (:65)
  constexpr int W = 2 * 40 + 1;                    // VITERBI_PATH_WIDTH = 40 (src/hhdecl.h:50)
  for (int w = threadIdx.x; w < ns * W; w += 256) {
    const int step = w / W, d = w - step * W - 40;
    const int i = pi[o + step], j = pj[o + step];
    if (i < 1 || i > g.Lq || j < 1 || j > Lt) continue;
    if (i + d >= 1 && i + d <= g.Lq) celloff_set(g, t, i + d, j);
    if (j + d >= 1 && j + d <= Lt) celloff_set(g, t, i, j + d);
  }
}

int celloff_from_paths(uint64_t* bt, const int64_t* rec_off, const int32_t* L, int64_t pass_stride, int Lq, StripPlan plan,
                       int n_templates, int n_paths, const int32_t* template_of, const int64_t* path_off, const int32_t* pi,
                       const int32_t* pj, const int32_t* ranges, int n_q, int n_t, hipStream_t stream) {
  CellOffGeom g{bt, rec_off, L, pass_stride, Lq, plan, 0};
  hipLaunchKernelGGL(celloff_clear_kernel, dim3(n_templates), dim3(256), 0, stream, g, ranges, n_q, n_t);
  if (n_paths > 0)
    hipLaunchKernelGGL(celloff_paths_kernel, dim3(n_paths), dim3(256), 0, stream, g, template_of, path_off, pi, pj);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int celloff_from_mask(uint64_t* bt, const int64_t* rec_off, const int32_t* L, int64_t pass_stride, int Lq, StripPlan plan, int t,
                      const unsigned char* d_mask, int Lt, hipStream_t stream) {
  CellOffGeom g{bt, rec_off, L, pass_stride, Lq, plan, 0};
  const int blocks = std::max(1, std::min(256, (Lt * plan.W + 255) / 256));
  hipLaunchKernelGGL(celloff_mask_kernel, dim3(blocks), dim3(256), 0, stream, g, t, Lt, d_mask);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

int bt_matrix(const uint64_t* bt, const int64_t* rec_off, int64_t pass_stride, int Lq, StripPlan plan, int bt_mm, int t, int Lt,
              unsigned char* d_out, hipStream_t stream) {
  CellOffGeom g{const_cast<uint64_t*>(bt), rec_off, nullptr, pass_stride, Lq, plan, bt_mm};
  const int64_t cells = (int64_t)(Lq + 1) * (Lt + 1);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(1024, (cells + 255) / 256));
  hipLaunchKernelGGL(bt_matrix_kernel, dim3(blocks), dim3(256), 0, stream, g, t, Lt, d_out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

}  // namespace hhv
