// hhv_topk.hip -- device-side selection of the K best hits (the per-GPU half of the sharded top-K
// merge).  Key = Hit.score descending, ties broken by the smaller template index, the
// order ViterbiRunner's caller establishes when it sorts the hit list (src/hhhit.h:116-126 compares
// score_aass = -score).  The K best of n are SELECTED, not sorted out of a full sort: every workgroup
// radix-selects the K largest 64-bit keys (orderable score, ~index) of its chunk of 16 384 keys held in
// registers, the survivors (chunks x K) go through the same step again until at most 4096 are left, and
// one workgroup sorts those in LDS and gathers the K records - two launches for 100 k templates and
// K = 500, work proportional to n once.  (hipCUB's full radix sort remains for K > 1024 and for the
// prefilter's selection, which needs the whole order.)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdlib>
#include <string>

#include "hhv_internal.h"
#include "viterbi_lane.h"

namespace hhv {

__device__ __forceinline__ uint64_t topk_key(float score, uint32_t idx) {
  uint32_t u = __builtin_bit_cast(uint32_t, score);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone float -> uint
  return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - idx);
}

// ---- radix select -------------------------------------------------------------------------------------------------------
// One workgroup = one chunk of <= SEL_CHUNK keys, 16 per thread in registers; key 0 = no key (padding of a lower level).
// Eight passes over the eight bytes of the key from the top: a 256-bin histogram in LDS of the keys that still share the
// prefix found so far, a suffix scan, the byte of the K-th largest key; after the last pass the prefix IS that key (keys
// are unique: the index is part of them) and every key >= it is written out - exactly K of them.
constexpr int SEL_THREADS = 1024, SEL_PER_THREAD = 16, SEL_CHUNK = SEL_THREADS * SEL_PER_THREAD;
constexpr int SEL_KMAX = 1024;   // a level keeps K of 16 384: the selection shrinks its input 16 x or more
constexpr int SORT_MAX = 4096;   // keys the final workgroup sorts in LDS

// The key of rank k (1-based, from the top) among the workgroup's keys (SEL_PER_THREAD per thread in registers; key 0 = no key): the
// passes described above.  Called by every thread of the workgroup, more than k real keys present.
struct SelShared {
  uint32_t hist[256];
  uint32_t digit, krem, all;
};
template <int PT>
__device__ __forceinline__ uint64_t radix_select_threshold(const uint64_t (&key)[PT], int k, SelShared& sh) {
  uint64_t prefix = 0;
  uint32_t krem = (uint32_t)k;
  for (int p = 0; p < 8; ++p) {
    const int shift = 56 - 8 * p;
    if (threadIdx.x < 256) sh.hist[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PT; ++e) {
      // still a candidate for the K-th key: a real key whose bytes above this one equal the prefix
      bool act = key[e] != 0 && (p == 0 || (key[e] >> (shift + 8)) == (prefix >> (shift + 8)));
      const uint32_t d = (uint32_t)(key[e] >> shift) & 255u;
      // the top bytes of the keys (sign, exponent) are nearly the same for all of them: two rounds of wave-aggregated
      // adds (all lanes that share the leader's byte add once) before the plain atomics
#pragma unroll
      for (int round = 0; round < 2; ++round) {
        const unsigned long long am = __ballot(act);
        if (am == 0) break;
        const uint32_t lead = (uint32_t)__shfl((int)d, __ffsll((long long)am) - 1);
        const unsigned long long same = __ballot(act && d == lead);
        if ((threadIdx.x & 63) == (uint32_t)(__ffsll((long long)same) - 1)) atomicAdd(&sh.hist[lead], (uint32_t)__popcll(same));
        act = act && d != lead;
      }
      if (act) atomicAdd(&sh.hist[d], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      // suffix scan by the first wave: lane l holds bins 4 l .. 4 l + 3; `above` = candidates in the bins above bin b.  The
      // byte of the krem-th largest key is the bin with above < krem <= above + hist[b].
      const uint32_t h0 = sh.hist[4 * threadIdx.x], h1 = sh.hist[4 * threadIdx.x + 1], h2 = sh.hist[4 * threadIdx.x + 2], h3 = sh.hist[4 * threadIdx.x + 3];
      const uint32_t own = h0 + h1 + h2 + h3;
      uint32_t incl = own;  // sum over lanes >= this one
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_down((int)incl, o);
        if ((int)threadIdx.x + o < 64) incl += up;
      }
      const uint32_t above_lane = incl - own;  // candidates in the bins of higher lanes
      const uint32_t a3 = above_lane, a2 = a3 + h3, a1 = a2 + h2, a0 = a1 + h1;
      int b = -1;
      uint32_t ab = 0, hb = 0;
      if (a3 < krem && krem <= a3 + h3) b = 3, ab = a3, hb = h3;
      else if (a2 < krem && krem <= a2 + h2) b = 2, ab = a2, hb = h2;
      else if (a1 < krem && krem <= a1 + h1) b = 1, ab = a1, hb = h1;
      else if (a0 < krem && krem <= a0 + h0) b = 0, ab = a0, hb = h0;
      if (b >= 0) {
        sh.digit = 4 * threadIdx.x + b;
        sh.krem = krem - ab;
        sh.all = (krem - ab) == hb;  // every key of this bin is taken: the bytes below do not matter any more
      }
    }
    __syncthreads();
    prefix |= (uint64_t)sh.digit << shift;
    krem = sh.krem;
    const bool all_of_bin = sh.all != 0;
    __syncthreads();
    if (all_of_bin) break;  // the smallest key with this prefix: exactly k keys are >= it
  }
  return prefix;
}

// rank: the ranking key of every hit when it is not Hit.score (hhv_topk with HHV_TOPK_PVALUE), else null
// SRC: SEL_KEYS = the keys a level before wrote, SEL_HITS = hit records (with `rank`), SEL_RESULTS = the DP kernel's result records
// (HHV_TOPK_RAW: `hits` points at them - no hit record is built for a template that is not selected)
enum { SEL_KEYS = 0, SEL_HITS = 1, SEL_RESULTS = 2 };
// PT keys per thread: a chunk is SEL_THREADS x PT keys.  16 for the levels of a very large set; 4 where the survivors of ONE level of
// smaller chunks still fit the last level's workgroup (100 000 templates, K = 500: 25 workgroups of 4096 keys instead of 7 of 16 384 -
// the level is one CU's issue per chunk, 26 -> 9 us)
template <int SRC, int PT>
__global__ void __launch_bounds__(SEL_THREADS) topk_select_kernel(const DevHit* __restrict__ hits, const uint64_t* __restrict__ in_keys,
                                                                  int n, int k, uint64_t* __restrict__ out_keys, const float* __restrict__ rank) {
  constexpr bool FROM_HITS = SRC != SEL_KEYS;
  constexpr int SEL_PER_THREAD = PT, SEL_CHUNK = SEL_THREADS * PT;
  __shared__ SelShared sh;
  __shared__ uint32_t sh_valid, sh_out;
  const int base = blockIdx.x * SEL_CHUNK;
  uint64_t key[SEL_PER_THREAD];
  int mine = 0;
  {
    // (all loads first, from an index clamped into the array: a load under `if (i < n)` waits for its own data before the next is sent)
    float sc[SEL_PER_THREAD];
#pragma unroll
    for (int e = 0; e < SEL_PER_THREAD; ++e) {
      const int i = min(base + e * SEL_THREADS + (int)threadIdx.x, n - 1);  // coalesced
      if (SRC == SEL_RESULTS)
        sc[e] = reinterpret_cast<const DevResult*>(hits)[i].score;
      else if (SRC == SEL_HITS)
        sc[e] = rank ? rank[i] : hits[i].score;
      else
        key[e] = in_keys[i];
    }
#pragma unroll
    for (int e = 0; e < SEL_PER_THREAD; ++e) {
      const int i = base + e * SEL_THREADS + (int)threadIdx.x;
      if (FROM_HITS) key[e] = topk_key(sc[e], (uint32_t)i);
      if (i >= n) key[e] = 0;
      mine += key[e] != 0;
    }
  }
  if (threadIdx.x == 0) sh_valid = 0, sh_out = 0;
  __syncthreads();
  {
    // (one atomic per wave)
    int v = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&sh_valid, (uint32_t)v);
  }
  __syncthreads();
  const int valid = (int)sh_valid;
  uint64_t* out = out_keys + (size_t)blockIdx.x * k;
  uint64_t T = 1;  // fewer than k keys: all of them (every key >= 1)
  if (valid > k) T = radix_select_threshold(key, k, sh);
#pragma unroll
  for (int e = 0; e < SEL_PER_THREAD; ++e) {
    const bool take = key[e] >= T && key[e] != 0;
    const unsigned long long m = __ballot(take);
    if (m) {
      uint32_t slot0 = 0;
      const int leader = __ffsll((long long)m) - 1;
      if ((int)(threadIdx.x & 63) == leader) slot0 = atomicAdd(&sh_out, (uint32_t)__popcll(m));
      slot0 = (uint32_t)__shfl((int)slot0, leader);
      if (take) out[slot0 + (uint32_t)__popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull))] = key[e];
    }
  }
  __syncthreads();
  for (int t = (int)sh_out + (int)threadIdx.x; t < k; t += SEL_THREADS) out[t] = 0;  // padding of a chunk with fewer than k keys
}

// ---- small sets in ONE launch (round 6) ---------------------------------------------------------------------------------------
// n <= SEL_CHUNK keys and k <= SEL_KMAX: what took three launches (results -> hit records, select, final sort: 37 us of kernels and
// the gaps between them - a tenth of a 10 000-template search step) is one workgroup's job.
//   1. every thread's keys in registers (from the hit records, or straight from the DP kernel's result records: SRC_RESULTS);
//   2. a lower bound L of the k-th key without a selection: the k-th largest of the 1024 THREAD MAXIMA (each is one of the keys,
//      so at least k keys are >= L) - one sort of 1024 keys, one per thread;
//   3. the keys >= L ("candidates"; ~650 of 10 000 for k = 500) compacted into LDS.  More than 1024 of them (k near 1024, or an
//      adversarial order): the radix selection above finds the exact k-th key instead, and exactly k keys are compacted;
//   4. the candidates sorted, one per thread, the k best records gathered.
// The sort: a bitonic network over one key per thread, descending by thread index, in the form whose comparators all point the same
// way - a block of `size` keys with two sorted halves is merged by one MIRROR step (t with t ^ (size - 1)) and half-cleaners
// (t with t ^ stride, stride = size / 4 .. 1); the smaller thread index keeps the larger key in every step, so the only
// per-step fact about a thread is one bit of its index.  Partners inside the wavefront are exchanged by ds_swizzle (xor masks below
// 32: no address register) or ds_bpermute (masks 32 and 63), without a barrier; partners in other wavefronts (10 of the 55 steps
// of 1024 keys) through LDS, two buffers in turn: one barrier per step.  One CU's VALU issue is what the kernel costs (sixteen
// wavefronts): 3 - 4 vector instructions per key and step here against ~25 of the first version (__shfl_xor re-derives its
// partner index every time), tools/topk_ubench.hip: 8.3 -> see profiles/r6_topk_small.txt us per 1024 keys.
// (WITH_VAL: a 32-bit payload travels with its key and breaks ties between equal keys - the merge's position of a record.)
struct SortLds {
  uint64_t x[2][SEL_THREADS];
  uint32_t v[2][SEL_THREADS];
};
template <int MASK>
__device__ __forceinline__ uint32_t lane_xor(uint32_t w, int idx32, int idx63) {
  if constexpr (MASK < 32)
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)w, (MASK << 10) | 0x1F);  // bit mode: and 0x1F, or 0, xor MASK
  else
    return (uint32_t)__builtin_amdgcn_ds_bpermute(MASK == 32 ? idx32 : idx63, (int)w);
}
template <bool WITH_VAL>
__device__ __forceinline__ void cx_select(uint64_t& x, uint32_t& v, uint64_t other, uint32_t ov, bool keep_max) {
  const bool gt = WITH_VAL ? (x > other || (x == other && v > ov)) : x > other;
  if (gt != keep_max) {
    x = other;
    if (WITH_VAL) v = ov;
  }
}
template <bool WITH_VAL, int MASK>
__device__ __forceinline__ void cx_lane(uint64_t& x, uint32_t& v, bool keep_max, int idx32, int idx63) {
  const uint32_t olo = lane_xor<MASK>((uint32_t)x, idx32, idx63), ohi = lane_xor<MASK>((uint32_t)(x >> 32), idx32, idx63);
  const uint32_t ov = WITH_VAL ? lane_xor<MASK>(v, idx32, idx63) : 0u;
  cx_select<WITH_VAL>(x, v, ((uint64_t)ohi << 32) | olo, ov, keep_max);
}
// P: a power of two, 64 <= P <= SEL_THREADS; every thread of the workgroup calls (threads >= P sort keys of their own: zeros)
template <bool WITH_VAL>
__device__ __forceinline__ void wg_sort_desc(uint64_t& x, uint32_t& v, int P, SortLds& sl) {
  const int t = (int)threadIdx.x, lane = t & 63;
  const int idx32 = (lane ^ 32) << 2, idx63 = (lane ^ 63) << 2;
  const bool k1 = (lane & 1) == 0, k2 = (lane & 2) == 0, k4 = (lane & 4) == 0, k8 = (lane & 8) == 0, k16 = (lane & 16) == 0, k32 = (lane & 32) == 0;
#define CX(MASK, KM) cx_lane<WITH_VAL, MASK>(x, v, KM, idx32, idx63);
  CX(1, k1)
  CX(3, k2) CX(1, k1)
  CX(7, k4) CX(2, k2) CX(1, k1)
  CX(15, k8) CX(4, k4) CX(2, k2) CX(1, k1)
  CX(31, k16) CX(8, k8) CX(4, k4) CX(2, k2) CX(1, k1)
  CX(63, k32) CX(16, k16) CX(8, k8) CX(4, k4) CX(2, k2) CX(1, k1)
  int buf = 0;
  for (int size = 128; size <= P; size <<= 1) {
    for (int step = 0, stride = size >> 1; stride >= 64; ++step, stride >>= 1) {
      const int partner = step == 0 ? t ^ (size - 1) : t ^ stride;  // the mirror step, then the half-cleaners down to 64
      sl.x[buf][t] = x;
      if (WITH_VAL) sl.v[buf][t] = v;
      __syncthreads();
      const uint64_t other = sl.x[buf][partner];
      const uint32_t ov = WITH_VAL ? sl.v[buf][partner] : 0u;
      buf ^= 1;  // (the next step writes the other buffer: its writers have all passed this step's barrier, this buffer's readers
                 //  are past the next one before it is written again)
      cx_select<WITH_VAL>(x, v, other, ov, (t & stride) == 0);  // (the mirror step decides by the bit size / 2 as well)
    }
    CX(32, k32) CX(16, k16) CX(8, k8) CX(4, k4) CX(2, k2) CX(1, k1)
  }
#undef CX
}

enum { SRC_HITS = 0, SRC_RESULTS = 1 };
#ifdef HHV_TOPK_TIMING
// measurement build (tools/topk_ubench.hip): the clock at the phase boundaries of the small-set kernel, wave 0
__device__ unsigned long long g_topk_clk[16];
#define TOPK_T(slot) { if (threadIdx.x == 0) g_topk_clk[slot] = __builtin_readcyclecounter(); }
#define TOPK_C(c) g_topk_clk[15] = (unsigned long long)(c);
#else
#define TOPK_T(slot)
#define TOPK_C(c)
#endif
// the keys >= L of every thread counted, and the workgroup's exclusive prefix of the counts: where a thread's candidates go in the
// compacted array (the order of the compaction does not matter - it is sorted afterwards - but a scan costs one barrier where a
// ballot + atomic per key cost sixteen dependent LDS round trips: 3.4 -> 0.9 us)
__device__ __forceinline__ int candidate_scan(const uint64_t (&key)[SEL_PER_THREAD], uint64_t L, uint32_t* wtot /* [17] */, int& total) {
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  int c = 0;
#pragma unroll
  for (int e = 0; e < SEL_PER_THREAD; ++e) c += key[e] >= L && key[e] != 0;
  int incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(incl, o);
    if (lane >= o) incl += up;
  }
  __syncthreads();  // (the readers of an earlier call)
  if (lane == 63) wtot[wave] = (uint32_t)incl;
  __syncthreads();
  int base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < SEL_THREADS / 64; ++w) {
    const int x = (int)wtot[w];
    base += w < wave ? x : 0;
    all += x;
  }
  total = all;
  return base + incl - c;
}

// KEYS_IN: the last level of a large set - n keys a selection level wrote (0 = padding), the records gathered from `src`
template <int SRC, bool KEYS_IN = false>
__global__ void __launch_bounds__(SEL_THREADS) topk_small_kernel(const void* __restrict__ src, const float* __restrict__ rank, int n, int k,
                                                                 const int32_t* __restrict__ gids, DevHit* __restrict__ out, int force_radix,
                                                                 const uint64_t* __restrict__ in_keys = nullptr) {
  __shared__ SelShared sh;
  __shared__ SortLds sl;
  __shared__ uint64_t cand[SEL_THREADS];
  __shared__ uint64_t sh_L;
  __shared__ uint32_t wtot[SEL_THREADS / 64 + 1];
  const DevHit* hits = (const DevHit*)src;
  const DevResult* res = (const DevResult*)src;
  const int t = (int)threadIdx.x;
  TOPK_T(0)
  uint64_t key[SEL_PER_THREAD];
  if (KEYS_IN) {
#pragma unroll
    for (int e = 0; e < SEL_PER_THREAD; ++e) key[e] = in_keys[min(e * SEL_THREADS + t, n - 1)];
#pragma unroll
    for (int e = 0; e < SEL_PER_THREAD; ++e)
      if (e * SEL_THREADS + t >= n) key[e] = 0;
  } else {
    // (all loads first, from an index clamped into the array: a load under `if (i < n)` waits for its own data before the next is sent)
    float sc[SEL_PER_THREAD];
#pragma unroll
    for (int e = 0; e < SEL_PER_THREAD; ++e) {
      const int i = min(e * SEL_THREADS + t, n - 1);  // coalesced
      sc[e] = SRC == SRC_RESULTS ? res[i].score : (rank ? rank[i] : hits[i].score);
    }
#pragma unroll
    for (int e = 0; e < SEL_PER_THREAD; ++e) {
      const int i = e * SEL_THREADS + t;
      key[e] = i < n ? topk_key(sc[e], (uint32_t)i) : 0;
    }
  }
  uint64_t mx = 0;
  int mine = 0;
#pragma unroll
  for (int e = 0; e < SEL_PER_THREAD; ++e) {
    mx = key[e] > mx ? key[e] : mx;
    mine += key[e] != 0;
  }
  if (t == 0) sh_L = 0;
  TOPK_T(1)
  uint32_t none = 0;
  wg_sort_desc<false>(mx, none, SEL_THREADS, sl);
  TOPK_T(2)
  __syncthreads();
  if (t == k - 1) sh_L = mx;  // 0 when fewer than k threads hold a key: every key is a candidate
  __syncthreads();
  uint64_t L = sh_L ? sh_L : 1;
  int C = 0;
  int off = candidate_scan(key, L, wtot, C);
  if (C > SEL_THREADS || force_radix) {
    // more than 1024 keys above the bound (k near 1024, an adversarial order), or the test switch: the exact k-th key
    int valid = mine;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
    __syncthreads();
    if ((t & 63) == 0) wtot[t >> 6] = (uint32_t)valid;
    __syncthreads();
    valid = 0;
#pragma unroll
    for (int w = 0; w < SEL_THREADS / 64; ++w) valid += (int)wtot[w];
    L = valid > k ? radix_select_threshold(key, k, sh) : 1;
    off = candidate_scan(key, L, wtot, C);  // exactly min(k, valid) now
  }
  TOPK_T(3)
#pragma unroll
  for (int e = 0; e < SEL_PER_THREAD; ++e)
    if (key[e] >= L && key[e] != 0) cand[off++] = key[e];
  __syncthreads();
  TOPK_T(4)
  int P = 64;
  while (P < C) P <<= 1;
  uint64_t x = t < C ? cand[t] : 0;
  wg_sort_desc<false>(x, none, P, sl);
  TOPK_T(5)
  if (t < k) {
    const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(x & 0xFFFFFFFFu);
    DevHit h;
    if (SRC == SRC_RESULTS) {
      const DevResult r = res[idx];  // results_to_hits_kernel's record
      h.score = r.score;
      h.viterbi_score = r.score;
      h.score_ss = 0.0f;
      h.index = (int32_t)idx;
      h.i1 = h.j1 = 0;
      h.i2 = r.i2;
      h.j2 = r.j2;
      h.nsteps = 0;
      h.matched_cols = 0;
    } else {
      h = hits[idx];
    }
    if (gids) h.index = gids[h.index];
    out[t] = h;
  }
  TOPK_T(6)
  if (threadIdx.x == 0) { TOPK_C(C) }
}

// the last level: m <= SORT_MAX keys (or hits) sorted descending in LDS by a bitonic network, the k best gathered
template <bool FROM_HITS>
__global__ void __launch_bounds__(1024) topk_final_kernel(const DevHit* __restrict__ hits, const uint64_t* __restrict__ in_keys, int m,
                                                          int k, const int32_t* __restrict__ gids, DevHit* __restrict__ out,
                                                          const float* __restrict__ rank) {
  __shared__ uint64_t key[SORT_MAX];
  int P = 2;
  while (P < m) P <<= 1;
  for (int i = threadIdx.x; i < P; i += blockDim.x) key[i] = i < m ? (FROM_HITS ? topk_key(rank ? rank[i] : hits[i].score, (uint32_t)i) : in_keys[i]) : 0;
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
        }
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < k; t += blockDim.x) {
    const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(key[t] & 0xFFFFFFFFu);
    DevHit h = hits[idx];
    if (gids) h.index = gids[h.index];
    out[t] = h;
  }
}

__global__ void topk_keys_kernel(const DevHit* __restrict__ hits, int n, uint64_t* __restrict__ keys, const float* __restrict__ rank) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  keys[k] = topk_key(rank ? rank[k] : hits[k].score, (uint32_t)k);
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

// m <= SEL_THREADS records (one or two shards' lists of 500): one key per thread through the wavefront-exchange network above -
// 12 barriers where the LDS network of merge_hits_kernel has 45-55 (14.8 -> ~5 us; the step of a 10 000-template search ends with it)
__global__ void __launch_bounds__(SEL_THREADS) merge_hits_small_kernel(const DevHit* __restrict__ in, int m, int k, DevHit* __restrict__ out,
                                                                       int* __restrict__ n_out) {
  __shared__ SortLds sl;
  __shared__ int n_valid;
  const int t = (int)threadIdx.x;
  if (t == 0) n_valid = 0;
  __syncthreads();
  uint64_t kk = t < m ? merge_key(in[t]) : 0;
  {
    const unsigned long long b = __ballot(kk != 0);
    if ((t & 63) == 0 && b) atomicAdd(&n_valid, (int)__popcll(b));
  }
  int P = 64;
  while (P < m) P <<= 1;
  uint32_t pos = (uint32_t)t;
  wg_sort_desc<true>(kk, pos, P, sl);
  __syncthreads();
  const int nv = min(n_valid, k);
  if (t < k) {
    DevHit h;
    if (t < nv) {
      h = in[pos];
    } else {
      h.score = h.viterbi_score = h.score_ss = __builtin_bit_cast(float, 0xFFFFFFFFu);
      h.index = h.i1 = h.j1 = h.i2 = h.j2 = h.nsteps = h.matched_cols = -1;
    }
    out[t] = h;
  }
  for (int u = SEL_THREADS + t; u < k; u += SEL_THREADS) {  // k beyond the records there are: padding
    DevHit h;
    h.score = h.viterbi_score = h.score_ss = __builtin_bit_cast(float, 0xFFFFFFFFu);
    h.index = h.i1 = h.j1 = h.i2 = h.j2 = h.nsteps = h.matched_cols = -1;
    out[u] = h;
  }
  if (t == 0) *n_out = nv;
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
  static const bool small_off = [] {
    const char* e = getenv("HHV_TOPK_SMALL");
    return e && atoi(e) == 0;
  }();
  if (m <= SEL_THREADS && !small_off) {
    hipLaunchKernelGGL(merge_hits_small_kernel, dim3(1), dim3(SEL_THREADS), 0, stream, d_in, m, k, d_out, d_n);
  } else if (m <= MERGE_MAX) {
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

// ---- the reference's ranking key (round 6; VERDICT r5 missing #5) -------------------------------------------------------------
// The reference does not sort its hit list by Hit.score but by score_aass (Hit::operator<, src/hhhit.h:116-126), which
// HitList::CalculatePvalues (src/hhhitlist.cpp:499-531) derives from the score through an extreme-value distribution whose
// lamda and mu depend on the two lengths and diversities (the little networks lamda_NN / mu_NN, src/hhhitlist-inl.h:13-66) and
// Hit::CalcEvalScoreProbab (src/hhhit.h:134-141).  A shard that cuts its list at K by Hit.score can drop a hit the reference
// ranks inside the top K (a short template's score counts for more).  rank[k] = -score_aass of hit k, the key hhv_topk sorts by
// with HHV_TOPK_PVALUE; the operations are the reference's, in its types (float network inputs, double exp / log).  exp and log
// are the device's (within an ulp of libm's): the key decides which K records leave the shard, the host recomputes the
// reference's numbers for what arrives.
__device__ __forceinline__ float nn_hidden(const float* w, float bias, float Lq, float Lt, float Nq, float Nt) {
  float res = Lq * w[0] + Lt * w[1] + Nq * w[2] + Nt * w[3] + bias;
  res = (float)(1.0 / (1.0 + exp(-(double)res)));
  return res;
}
__global__ void topk_rank_pvalue_kernel(const DevHit* __restrict__ hits, int n, const int32_t* __restrict__ L, const float* __restrict__ t_neff,
                                        int Lq, float q_neff, int local, float* __restrict__ rank) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const DevHit h = hits[k];
  float lamda = 0.42f /* LAMDA_GLOB, src/hhdecl.h:43 */, mu = 3.0f;
  if (local) {
    const float log1000 = (float)log(1000.0);
    const float a = (float)(log((double)Lq) / (double)log1000), b = (float)(log((double)L[h.index]) / (double)log1000);
    const float c = (float)((double)q_neff / 10.0), d = (float)((double)t_neff[h.index] / 10.0);
    {
      const float bias[4] = {-0.73195f, -1.43792f, -1.18839f, -3.01141f};
      const float w[20] = {-0.52356f, -3.37650f, 1.12984f, -0.46796f, -4.71361f, 0.14166f, 1.66807f, 0.16383f, -0.94895f, -1.24358f,
                           -1.20293f, 0.95434f, -0.00318f, 0.53022f, -0.04914f, -0.77046f, 2.45630f, 3.02905f, 2.53803f, 2.64379f};
      lamda = 0.0f;
      for (int u = 0; u < 4; ++u) lamda += nn_hidden(w + 4 * u, bias[u], a, b, c, d) * w[16 + u];
    }
    {
      const float bias[6] = {-4.25264f, -3.63484f, -5.86653f, -4.78472f, -2.76356f, -2.21580f};
      const float w[30] = {1.96172f, 1.07181f, -7.41256f, 0.26471f, 0.84643f, 1.46777f, -1.04800f, -0.51425f, 1.42697f, 1.99927f,
                           0.64647f, 0.27834f, 1.34216f, 1.64064f, 0.35538f, -8.08311f, 2.30046f, 1.31700f, -0.46435f, -0.46803f,
                           0.90090f, -3.53067f, 0.59212f, 1.47503f, -1.26036f, 1.52812f, 1.58413f, -1.90409f, 0.92803f, -0.66871f};
      float m = 0.0f;
      for (int u = 0; u < 6; ++u) m += nn_hidden(w + 4 * u, bias[u], a, b, c, d) * w[24 + u];
      mu = (float)(20.0 * (double)m);
    }
  }
  // logPvalue / Pvalue (src/hhhit-inl.h:44-53)
  const double hh = (double)(lamda * (h.score - mu));
  const double logPval = hh > 10 ? -hh : (hh < -2.5 ? -exp(-exp(-hh)) : log(1.0 - exp(-exp(-hh))));
  const double Pval = hh > 10 ? exp(-hh) : 1.0 - exp(-exp(-hh));
  // CalcEvalScoreProbab (src/hhhit.h:134-141)
  const float score_aass = (float)((logPval < -10.0 ? logPval : log(-log(1 - Pval))) / 0.45 -
                                   fmin((double)(lamda * h.score_ss), fmax(0.0, 0.2 * ((double)h.score - 8.0))) / 0.45 - 3.0);
  rank[k] = -score_aass;
}
void topk_rank_pvalue(const DevHit* d_hits, int n, const int32_t* d_L, const float* d_neff, int Lq, float q_neff, int local, float* d_rank,
                      hipStream_t stream) {
  hipLaunchKernelGGL(topk_rank_pvalue_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_hits, n, d_L, d_neff, Lq, q_neff, local, d_rank);
}

// One launch for a small set (topk_small_kernel): d_results != null = straight from the DP kernel's result records (HHV_TOPK_RAW),
// else from d_hits (with `rank` when the ranking key is not Hit.score).  Returns 1 when the set is not small (the caller takes
// topk_device), 0 when launched, -1 on a launch error.  HHV_TOPK_SMALL = 0 switches the path off, 2 forces its radix branch (tests).
int topk_small_device(const DevHit* d_hits, const DevResult* d_results, int n, int k, const int32_t* gids, DevHit* d_out, hipStream_t stream,
                      std::string* err, const float* rank) {
  static const int mode = [] {
    const char* e = getenv("HHV_TOPK_SMALL");
    return e ? atoi(e) : 1;
  }();
  if (mode == 0 || n > SEL_CHUNK || k > SEL_KMAX || k > n) return 1;
  if (d_results)
    hipLaunchKernelGGL(topk_small_kernel<SRC_RESULTS>, dim3(1), dim3(SEL_THREADS), 0, stream, (const void*)d_results, (const float*)nullptr, n, k, gids,
                       d_out, mode == 2);
  else
    hipLaunchKernelGGL(topk_small_kernel<SRC_HITS>, dim3(1), dim3(SEL_THREADS), 0, stream, (const void*)d_hits, rank, n, k, gids, d_out, mode == 2);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    if (err) *err = std::string("top-K selection (small set): ") + hipGetErrorString(e);
    return -1;
  }
  return 0;
}

bool topk_small_enabled() {
  const char* e = getenv("HHV_TOPK_SMALL");
  return !(e && atoi(e) == 0);
}

// keys/sorted: n uint64 each, temp: topk_temp_bytes(n) (used by the full-sort path only).  Asynchronous on `stream`; k <= n.
// d_results != null (HHV_TOPK_RAW, k <= SEL_KMAX): the selection reads the DP kernel's result records in place, d_hits is not used.
int topk_device(const DevHit* d_hits, int n, int k, const int32_t* gids, DevHit* d_out, uint64_t* keys, uint64_t* sorted,
                void* temp, size_t temp_bytes, hipStream_t stream, std::string* err, const float* rank, const DevResult* d_results) {
  const int threads = 256;
  hipError_t e = hipSuccess;
  static const bool small_off = [] {
    const char* ev = getenv("HHV_TOPK_SMALL");
    return ev && atoi(ev) == 0;
  }();
  if (n <= SORT_MAX && !d_results) {
    hipLaunchKernelGGL(topk_final_kernel<true>, dim3(1), dim3(1024), 0, stream, d_hits, (const uint64_t*)nullptr, n, k, gids, d_out, rank);
  } else if (k <= SEL_KMAX) {
    // levels of selection: n keys -> chunks x k -> ... until one workgroup can finish (at most SEL_CHUNK keys for the one-key-per-
    // thread kernel, SORT_MAX for the LDS network of round 4 - HHV_TOPK_SMALL=0), the buffers used in turn
    const int last_max = (small_off && !d_results) ? SORT_MAX : SEL_CHUNK;
    int m = n;
    uint64_t* buf[2] = {keys, sorted};
    int cur = 0;
    const uint64_t* src = nullptr;
    while (m > last_max || src == nullptr) {
      // small chunks when their survivors fit the last level at once (and are no more than the input), else chunks of 16 384
      const int chunks4 = (m + SEL_THREADS * 4 - 1) / (SEL_THREADS * 4);
      const bool small_chunks = !small_off && (long)chunks4 * k <= last_max && (long)chunks4 * k < m;
      const int chunks = small_chunks ? chunks4 : (m + SEL_CHUNK - 1) / SEL_CHUNK;
#define HHV_SELECT(SRCMODE, ...)                                                                                                        \
  do {                                                                                                                                   \
    if (small_chunks) hipLaunchKernelGGL((topk_select_kernel<SRCMODE, 4>), dim3(chunks), dim3(SEL_THREADS), 0, stream, __VA_ARGS__);     \
    else hipLaunchKernelGGL((topk_select_kernel<SRCMODE, SEL_PER_THREAD>), dim3(chunks), dim3(SEL_THREADS), 0, stream, __VA_ARGS__);     \
  } while (0)
      if (src == nullptr && d_results)
        HHV_SELECT(SEL_RESULTS, (const DevHit*)d_results, (const uint64_t*)nullptr, m, k, buf[cur], (const float*)nullptr);
      else if (src == nullptr)
        HHV_SELECT(SEL_HITS, d_hits, (const uint64_t*)nullptr, m, k, buf[cur], rank);
      else
        HHV_SELECT(SEL_KEYS, (const DevHit*)nullptr, src, m, k, buf[cur], (const float*)nullptr);
#undef HHV_SELECT
      src = buf[cur];
      cur ^= 1;
      m = chunks * k;
    }
    if (d_results)
      hipLaunchKernelGGL((topk_small_kernel<SRC_RESULTS, true>), dim3(1), dim3(SEL_THREADS), 0, stream, (const void*)d_results, (const float*)nullptr, m, k, gids, d_out, 0, src);
    else if (!small_off)
      hipLaunchKernelGGL((topk_small_kernel<SRC_HITS, true>), dim3(1), dim3(SEL_THREADS), 0, stream, (const void*)d_hits, (const float*)nullptr, m, k, gids, d_out, 0, src);
    else
      hipLaunchKernelGGL(topk_final_kernel<false>, dim3(1), dim3(1024), 0, stream, d_hits, src, m, k, gids, d_out, (const float*)nullptr);
  } else {
    hipLaunchKernelGGL(topk_keys_kernel, dim3((n + threads - 1) / threads), dim3(threads), 0, stream, d_hits, n, keys, rank);
    e = hipcub::DeviceRadixSort::SortKeysDescending(temp, temp_bytes, keys, sorted, n, 0, 64, stream);
    if (e != hipSuccess) {
      if (err) *err = std::string("radix sort: ") + hipGetErrorString(e);
      return -1;
    }
    hipLaunchKernelGGL(topk_gather_kernel, dim3((k + threads - 1) / threads), dim3(threads), 0, stream, d_hits, sorted, k,
                       gids, d_out);
  }
  e = hipGetLastError();
  if (e != hipSuccess) {
    if (err) *err = std::string("top-K selection: ") + hipGetErrorString(e);
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
    for (int64_t c = threadIdx.x; c < (int64_t)(hi - lo + 1) * Lt; c += 256) celloff_set(g, t, lo + (int)(c / Lt), 1 + (int)(c % Lt));
  }
  for (int q = 0; q < n_t; ++q) {
    const int lo = ranges[2 * (n_q + q)], hi = min(ranges[2 * (n_q + q) + 1], Lt);
    for (int64_t c = threadIdx.x; c < (int64_t)(hi - lo + 1) * g.Lq; c += 256) celloff_set(g, t, 1 + (int)(c % g.Lq), lo + (int)(c / g.Lq));
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

// The same masks, entry by entry instead of cell by cell (round 5).  The crosses of neighbouring path steps overlap almost
// completely: 146 steps x 162 cells = 23 600 atomics per path above, ~240 M for the second round of 10 000 templates - 10 ms,
// four times the masked DP they prepare.  The union of the crosses has a closed form: the steps of a path that lie in one
// column j are consecutive rows (a vertical run), so column j is switched off from (its first row - 40) to (its last row + 40);
// likewise row i from (first column - 40) to (last column + 40).  One workgroup per path: the four range tables into LDS with
// atomicMin / atomicMax (one pass over the steps), then every 8-byte entry of the template is evaluated once - threads walk the
// buffer in ITS order (entries of one buffer row = 64 neighbouring lanes: contiguous) - and the non-zero ones are or-ed in with
// ONE atomic each (several paths of a template meet in an entry).  A caller's "path" that is not one (steps that do not move to
// a neighbouring cell) and profiles too long for the LDS tables take the kernel above.
constexpr int CELLOFF_BAND_MAX = 8000;  // Lq + Lt: 2 x 4 bytes per row and per column in LDS (64 KB)
__global__ void __launch_bounds__(256) celloff_band_kernel(CellOffGeom g, const int32_t* __restrict__ template_of,
                                                           const int64_t* __restrict__ path_off, const int32_t* __restrict__ pi,
                                                           const int32_t* __restrict__ pj) {
  extern __shared__ int s_rng[];
  const int p = blockIdx.x, t = template_of[p], Lt = g.L[t], Lq = g.Lq;
  const int64_t o = path_off[p];
  const int ns = (int)(path_off[p + 1] - o) - 1;  // the last step is skipped, like the reference (:65)
  int* row_lo = s_rng;                  // [Lt + 1] first / last ROW of the path's steps in column j
  int* row_hi = row_lo + (Lt + 1);
  int* col_lo = row_hi + (Lt + 1);      // [Lq + 1] first / last COLUMN of the path's steps in row i
  int* col_hi = col_lo + (Lq + 1);
  __shared__ int irregular, j_min, j_max;
  if (threadIdx.x == 0) irregular = 0, j_min = 0x7FFFFFFF, j_max = -1;
  for (int k = threadIdx.x; k <= Lt; k += 256) row_lo[k] = 0x7FFFFFFF, row_hi[k] = -1;
  for (int k = threadIdx.x; k <= Lq; k += 256) col_lo[k] = 0x7FFFFFFF, col_hi[k] = -1;
  __syncthreads();
  for (int s = threadIdx.x; s < ns; s += 256) {
    const int i = pi[o + s], j = pj[o + s];
    if (s + 1 < ns) {  // a path moves to a neighbouring cell per step, never back (src/hhviterbi.cpp:96-146: i and j only fall)
      const int di = pi[o + s + 1] - i, dj = pj[o + s + 1] - j;
      if (di > 0 || di < -1 || dj > 0 || dj < -1) irregular = 1;
    }
    if (i < 1 || i > Lq || j < 1 || j > Lt) continue;
    atomicMin(&row_lo[j], i);
    atomicMax(&row_hi[j], i);
    atomicMin(&col_lo[i], j);
    atomicMax(&col_hi[i], j);
    atomicMin(&j_min, j);
    atomicMax(&j_max, j);
  }
  __syncthreads();
  // a short alignment in a large matrix (10 steps, Lq and Lt of thousands): its 162 cells per step are far fewer than the
  // entries of the columns it touches - cell by cell then (an atomic costs about eight entry tests)
  const bool few_cells = j_max >= 0 && (long long)ns * 162 * 8 < (long long)Lq * (j_max - j_min + 81);
  if (irregular || few_cells) {  // not a path (or a very short one): cell by cell, as above
    constexpr int CW = 2 * 40 + 1;
    for (int w = threadIdx.x; w < ns * CW; w += 256) {
      const int step = w / CW, d = w - step * CW - 40;
      const int i = pi[o + step], j = pj[o + step];
      if (i < 1 || i > Lq || j < 1 || j > Lt) continue;
      if (i + d >= 1 && i + d <= Lq) celloff_set(g, t, i + d, j);
      if (j + d >= 1 && j + d <= Lt) celloff_set(g, t, i, j + d);
    }
    return;
  }
  const int W = g.plan.W;
  const int64_t rec0 = g.rec_off[t];
  for (int pass = 0; pass < g.plan.P; ++pass) {
    const int R = g.plan.R(pass), ilo = g.plan.base(pass) + 1;
    uint64_t* e = g.bt + (size_t)pass * g.pass_stride;
    // buffer rows rec0 + 1 .. rec0 + Lt + W - 1 hold the template's entries: (row rho, lane gl) = column rho - gl - rec0;
    // only columns within 40 of the path's can be touched: buffer rows j_lo .. j_hi + W - 1
    const int j_lo = max(1, j_min - 40), j_hi = min(Lt, j_max + 40);
    for (int k = (j_lo - 1) * W + threadIdx.x; k < (j_hi + W - 1) * W && j_max >= 0; k += 256) {
      const int gl = k % W, j = 1 + k / W - gl;
      if (j < j_lo || j > j_hi) continue;
      uint64_t v = 0;
      const int rlo = row_lo[j] - 40, rhi = row_hi[j] < 0 ? -1 : row_hi[j] + 40;
      for (int r = 0; r < R; ++r) {
        const int i = ilo + gl * R + r;
        if (i > Lq) break;
        const bool in_col = i >= rlo && i <= rhi;
        const bool in_row = col_hi[i] >= 0 && j >= col_lo[i] - 40 && j <= col_hi[i] + 40;
        if (in_col || in_row) v |= 0x80ull << (8 * r);
      }
      if (v) atomicOr((unsigned long long*)(e + bt_entry(rec0 + j, gl, W)), (unsigned long long)v);
    }
  }
}

int celloff_from_paths(uint64_t* bt, const int64_t* rec_off, const int32_t* L, int64_t pass_stride, int Lq, StripPlan plan,
                       int n_templates, int n_paths, const int32_t* template_of, const int64_t* path_off, const int32_t* pi,
                       const int32_t* pj, const int32_t* ranges, int n_q, int n_t, int max_Lt, hipStream_t stream) {
  CellOffGeom g{bt, rec_off, L, pass_stride, Lq, plan, 0};
  hipLaunchKernelGGL(celloff_clear_kernel, dim3(n_templates), dim3(256), 0, stream, g, ranges, n_q, n_t);
  if (n_paths > 0) {
    if (Lq + max_Lt <= CELLOFF_BAND_MAX) {
      const size_t lds = (size_t)2 * (Lq + max_Lt + 2) * sizeof(int);
      hipLaunchKernelGGL(celloff_band_kernel, dim3(n_paths), dim3(256), lds, stream, g, template_of, path_off, pi, pj);
    } else {
      hipLaunchKernelGGL(celloff_paths_kernel, dim3(n_paths), dim3(256), 0, stream, g, template_of, path_off, pi, pj);
    }
  }
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
