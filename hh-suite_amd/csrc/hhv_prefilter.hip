// hhv_prefilter.hip -- HHblits prefilter kernels on MI355X: replaces
//   Prefilter::ungapped_sse_score   src/hhprefilter.cpp:214-278   gapless profile-vs-sequence score
//   Prefilter::swStripedByte        src/hhprefilter.cpp:70-212    striped Smith-Waterman (Farrar / Zhao)
// for one query column-state profile against a resident database of column-state sequences.
//
// The reference vectorises along the QUERY with 32 unsigned bytes per AVX2 register ("striped": vector
// element k of segment row j is query position k*W + j, W = ceil(Lq/32)).
//
//  * Smith-Waterman: a 32-lane half-wavefront IS that vector.  Lane k owns stripe element k and keeps its W cells
//    of H and E in VGPRs (W is a template parameter, the j loop is unrolled), so the recurrence - including what
//    depends on the striping: E is updated before the lazy-F correction, the F chain restarts per segment in the
//    main loop - is reproduced by construction with 32-bit lanes doing the saturating uint8 arithmetic.  The
//    register shift of the reference (simdi8_shiftl) is one DPP wave_shr.
//  * The gapless score max_ij S(i,j), S(i,j) = sat(S(i-1,j-1) + q(i,x_j) - offset), does not depend on the striping
//    at all (each diagonal is an independent saturating chain), so a whole 64-lane wavefront takes one sequence,
//    W = ceil(Lq/64) cells per lane in VGPRs, the residue x_j is wave-uniform (scalar loads).
//
// Both fast kernels keep the query profile in LDS as SIGNED bytes q - offset, lane-major ([state][lane][W padded to
// 4]), so one ds_read_b32/b64/b128 fetches all W operands of a lane and each cell is add(SDWA byte) + med3 (+ max).
// They need q - offset to fit int8 and the profile to fit LDS (Lq <= 512); anything else takes the generic kernel
// (state in LDS, profile in LDS or global), which is the direct transcription of the AVX2 loops.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "hhv_internal.h"

namespace hhv {

// clamp(x, 0, cap) with a wave-uniform cap: one VOP3 (cap rides the constant bus)
__device__ __forceinline__ int med3i(int x, int cap) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(cap));
  return r;
}
__device__ __forceinline__ int sat_sub(int a, int b) { return (int)__builtin_elementwise_sub_sat((unsigned)a, (unsigned)b); }
__device__ __forceinline__ int sbyte(uint32_t w, int b) { return (int)(signed char)(w >> (8 * b)); }

// whole-wave shift by one lane, lane 0 <- 0
__device__ __forceinline__ int wave_shr1_zero(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
}
// simdi8_shiftl(x, 1) on a 32-lane half: element k <- element k-1, element 0 <- 0
__device__ __forceinline__ int half_shr1_zero(int v, int k) {
  const int up = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);
  return k == 0 ? 0 : up;
}

// profile -> LDS, lane-major signed bytes: word[(x * LANES + k) * WQ + q4] holds cells t = 4*q4 .. 4*q4+3 of lane k,
// cell t = query position k*W + t; padding cells hold 0 (= the reference's padding value `offset`, minus offset)
template <int LANES, int W>
__device__ __forceinline__ void fill_profile_lds(uint32_t* sprof, const PrefilterArgs& a, int nthreads, int q_base = 0) {
  constexpr int WQ = (W + 3) / 4;
  for (int e = threadIdx.x; e < 220 * LANES * WQ; e += nthreads) {
    const int q4 = e % WQ, k = (e / WQ) % LANES, x = e / (WQ * LANES);
    uint32_t w = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int t = q4 * 4 + b, pos = q_base + k * W + t;
      const int v = (t < W && pos < a.Lq) ? (int)a.profile[(size_t)x * a.Lq + pos] - a.offset : 0;
      w |= (uint32_t)(v & 0xff) << (8 * b);
    }
    sprof[e] = w;
  }
}

template <int WQ>
__device__ __forceinline__ void read_cells(const uint32_t* p, uint32_t (&w)[WQ]) {
  if (WQ == 5) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    w[0] = v.x;
    w[1] = v.y;
    w[2] = v.z;
    w[3] = v.w;
    w[4] = p[4];
  } else if (WQ == 1) {
    w[0] = p[0];
  } else if (WQ == 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    w[0] = v.x;
    w[1] = v.y;
  } else if (WQ == 3) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    w[0] = v.x;
    w[1] = v.y;
    w[2] = p[2];
  } else {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    w[0] = v.x;
    w[1] = v.y;
    w[2] = v.z;
    w[3] = v.w;
  }
}

// bytes lo..hi-1 of a sequence word are residues, the others belong to the neighbours / the padding: they become the
// null state 220+1 = PF_NULL whose profile row is all zero (shifts the diagonals, never raises a score)
constexpr int PF_NULL = 220;
__device__ __forceinline__ uint32_t mask_word(uint32_t word, int lo, int hi) {
  const uint32_t keep = (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
  return (word & keep) | (0xDCDCDCDCu & ~keep);
}

// ---- gapless score: one wavefront per sequence ----------------------------------------------------------------
template <int W, bool SLAB>
__global__ void __launch_bounds__(1024) hhv_pf_ungapped_kernel(PrefilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WQ = (W + 3) / 4;
  uint32_t* sprof = reinterpret_cast<uint32_t*>(smem);
  // a query longer than 64*W positions is processed in slabs of 64*W rows (one launch per slab): the diagonals are
  // carried from slab to slab through one byte per residue (the S value of the slab's last row), which is exact -
  // the gapless score does not depend on how the query is cut
  fill_profile_lds<64, W>(sprof, a, 1024, SLAB ? a.q_base : 0);
  for (int e = threadIdx.x; e < 64 * WQ; e += 1024) sprof[PF_NULL * 64 * WQ + e] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int cap = 255 - a.offset;
  const int wave = blockIdx.x * 16 + (threadIdx.x >> 6), n_waves = gridDim.x * 16;
  const uint32_t* mine = sprof + lane * WQ;
  const uint32_t* words = reinterpret_cast<const uint32_t*>(a.seqs);

  for (int64_t job = wave; job < a.n_jobs; job += n_waves) {
    const int slot = __builtin_amdgcn_readfirstlane(a.order ? a.order[job] : (int)job);
    const int sid = __builtin_amdgcn_readfirstlane(a.subset ? a.subset[slot] : slot);
    const int64_t beg = a.offsets[sid], end = a.offsets[sid + 1];
    // the sequence as aligned dwords w0 .. w0+nw-1; lane l keeps dword (chunk*64 + l), the next chunk is in flight
    const int64_t w0 = beg >> 2;
    const int nw = __builtin_amdgcn_readfirstlane((int)(((end + 3) >> 2) - w0));
    const int head = __builtin_amdgcn_readfirstlane((int)(beg & 3));
    const int tail = __builtin_amdgcn_readfirstlane((int)(end - ((w0 + nw - 1) << 2)));  // residues in the last word
    int S[W];
#pragma unroll
    for (int t = 0; t < W; ++t) S[t] = 0;
    int vmax = 0;
    const uint32_t* cin = SLAB ? reinterpret_cast<const uint32_t*>(a.carry_in) : nullptr;
    int c_prev = 0;  // S(last row of the previous slab, j-1); 0 in front of the sequence
    uint32_t chunk_next = words[w0 + max(min(lane, nw - 1), 0)];
    uint32_t cchunk_next = cin ? cin[w0 + max(min(lane, nw - 1), 0)] : 0u;
    for (int c0 = 0; c0 < nw; c0 += 64) {
      const uint32_t chunk = chunk_next, cchunk = cchunk_next;
      chunk_next = words[w0 + min(c0 + 64 + lane, nw - 1)];
      if (cin) cchunk_next = cin[w0 + min(c0 + 64 + lane, nw - 1)];
      const int n_here = min(64, nw - c0);
      for (int wl = 0; wl < n_here; ++wl) {
        uint32_t word = __builtin_amdgcn_readlane(chunk, wl);
        uint32_t cword = cin ? (uint32_t)__builtin_amdgcn_readlane(cchunk, wl) : 0u;
        const int wi = c0 + wl;
        if (wi == 0 || wi == nw - 1) {
          const int lo = wi == 0 ? head : 0, hi = wi == nw - 1 ? tail : 4;
          word = mask_word(word, lo, hi);
          // carry bytes outside the sequence belong to its neighbours: no carry there
          cword &= (hi >= 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u)) & ~((1u << (8 * lo)) - 1u);
        }
        uint32_t p[4][WQ];
#pragma unroll
        for (int b = 0; b < 4; ++b) read_cells<WQ>(mine + ((word >> (8 * b)) & 0xff) * (64 * WQ), p[b]);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int carry = wave_shr1_zero(S[W - 1]);
          if (cin) {
            carry = lane == 0 ? c_prev : carry;
            c_prev = (cword >> (8 * b)) & 0xff;
          }
#pragma unroll
          for (int t = W - 1; t >= 1; --t) S[t] = med3i(S[t - 1] + sbyte(p[b][t >> 2], t & 3), cap);
          S[0] = med3i(carry + sbyte(p[b][0], 0), cap);
#pragma unroll
          for (int t = 0; t < W; ++t) vmax = max(vmax, S[t]);
          if (SLAB && a.carry_out) {
            const int64_t pos = (w0 + wi) * 4 + b;
            if (lane == 63 && pos >= beg && pos < end) a.carry_out[pos] = (unsigned char)S[W - 1];
          }
        }
      }
    }
    for (int o = 32; o >= 1; o >>= 1) vmax = max(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0) a.scores[slot] = (SLAB && a.q_base) ? max(vmax, a.scores[slot]) : vmax;
  }
}

// ---- gapless score, TWO cells per register (round 6, queries up to 320 columns) ---------------------------------------
// The cells of a lane as pairs of int16 - (S[2p], S[2p+1]) in one VGPR - and the profile as int16 pairs in LDS ([state][lane][W / 2]
// words, the odd cell of W = 1, 3, 5 as a halfword array beside them: 221 x 64 x 10 B = 141 KB at W = 5).  A residue costs per pair
// one v_alignbit (the pair shifted by one cell: (S[2p-1], S[2p])), one v_pk_add_i16 with clamp, one v_pk_max_i16 against 0 and one
// v_pk_max_i16 into the running maximum: 4 instructions for two cells where the byte kernel above spends 3 per cell (add with a
// byte select, med3, max) - on paper; with the shift, the carry and the address arithmetic it is 16 against 17 per residue at
// W = 5, and slower in practice (see launch_prefilter_fast).  The UPPER clamp of the reference's saturating byte arithmetic (255 - offset) is not applied per cell:
// a diagonal's values equal the clamped ones until the first of them passes the cap, at that cell the clamped chain holds
// exactly the cap, so max over all cells, cut at the cap at the very end, is the same number; int16 saturation keeps an
// unclamped chain from wrapping.  The slot beside the odd cell of an odd W carries a copy of that cell's previous value (its
// profile operand is 0): never above the running maximum.
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_add_sat(uint32_t x, uint32_t y) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(i16x2, x), __builtin_bit_cast(i16x2, y)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t x, uint32_t y) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, x), __builtin_bit_cast(i16x2, y)));
}
template <int W>
__global__ void __launch_bounds__(1024) hhv_pf_ungapped_pk_kernel(PrefilterArgs a) {
  constexpr int NP = W / 2, ODD = W & 1, NPR = NP + ODD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* spair = reinterpret_cast<uint32_t*>(smem);                                   // [221][64][NP]
  uint16_t* sodd = reinterpret_cast<uint16_t*>(spair + (size_t)(PF_NULL + 1) * 64 * NP);  // [221][64], odd W only
  auto cell = [&](int x, int k, int t) -> int {
    const int pos = k * W + t;
    return (x < PF_NULL && t < W && pos < a.Lq) ? (int)a.profile[(size_t)x * a.Lq + pos] - a.offset : 0;
  };
  constexpr int NPD = NP > 0 ? NP : 1;
  for (int e = threadIdx.x; e < (PF_NULL + 1) * 64 * NP; e += 1024) {
    const int p = e % NPD, k = (e / NPD) % 64, x = e / (NPD * 64);
    spair[e] = ((uint32_t)cell(x, k, 2 * p) & 0xffffu) | ((uint32_t)cell(x, k, 2 * p + 1) << 16);
  }
  if (ODD)
    for (int e = threadIdx.x; e < (PF_NULL + 1) * 64; e += 1024) sodd[e] = (uint16_t)cell(e / 64, e % 64, W - 1);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int cap = 255 - a.offset;
  const int wave = blockIdx.x * 16 + (threadIdx.x >> 6), n_waves = gridDim.x * 16;
  const uint32_t* words = reinterpret_cast<const uint32_t*>(a.seqs);

  for (int64_t job = wave; job < a.n_jobs; job += n_waves) {
    const int slot = __builtin_amdgcn_readfirstlane(a.order ? a.order[job] : (int)job);
    const int sid = __builtin_amdgcn_readfirstlane(a.subset ? a.subset[slot] : slot);
    const int64_t beg = a.offsets[sid], end = a.offsets[sid + 1];
    const int64_t w0 = beg >> 2;
    const int nw = __builtin_amdgcn_readfirstlane((int)(((end + 3) >> 2) - w0));
    const int head = __builtin_amdgcn_readfirstlane((int)(beg & 3));
    const int tail = __builtin_amdgcn_readfirstlane((int)(end - ((w0 + nw - 1) << 2)));  // residues in the last word
    uint32_t P[NPR], vm = 0;
#pragma unroll
    for (int p = 0; p < NPR; ++p) P[p] = 0;
    uint32_t chunk_next = words[w0 + max(min(lane, nw - 1), 0)];
    for (int c0 = 0; c0 < nw; c0 += 64) {
      const uint32_t chunk = chunk_next;
      chunk_next = words[w0 + min(c0 + 64 + lane, nw - 1)];
      const int n_here = min(64, nw - c0);
      for (int wl = 0; wl < n_here; ++wl) {
        uint32_t word = __builtin_amdgcn_readlane(chunk, wl);
        const int wi = c0 + wl;
        if (wi == 0 || wi == nw - 1) word = mask_word(word, wi == 0 ? head : 0, wi == nw - 1 ? tail : 4);
        uint32_t q[4][NPR];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int x = (word >> (8 * b)) & 0xff;
          if (NP > 0) {
            uint32_t t[NP > 0 ? NP : 1];
            read_cells<(NP > 0 ? NP : 1)>(spair + ((size_t)x * 64 + lane) * NP, t);
#pragma unroll
            for (int p = 0; p < NP; ++p) q[b][p] = t[p];
          }
          if (ODD) q[b][NPR - 1] = sodd[x * 64 + lane];  // (q, 0)
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          // the last cell of the lane on the left: low half of its last pair (odd W) or high half (even W); lane 0: row 0's zero
          const uint32_t sl = (uint32_t)wave_shr1_zero((int)P[NPR - 1]);
          uint32_t sh[NPR];
#pragma unroll
          for (int p = NPR - 1; p >= 1; --p) sh[p] = __builtin_amdgcn_alignbit(P[p], P[p - 1], 16);  // (S[2p-1], S[2p])
          sh[0] = ODD ? __builtin_amdgcn_perm(P[0], sl, 0x05040100u) : __builtin_amdgcn_alignbit(P[0], sl, 16);  // (carry, S[0])
#pragma unroll
          for (int p = 0; p < NPR; ++p) {
            P[p] = pk_max(pk_add_sat(sh[p], q[b][p]), 0u);
            vm = pk_max(vm, P[p]);
          }
        }
      }
    }
    int vmax = max((int)(short)(vm & 0xffffu), (int)(short)(vm >> 16));
    for (int o = 32; o >= 1; o >>= 1) vmax = max(vmax, __shfl_xor(vmax, o, 64));
    if (lane == 0) a.scores[slot] = min(vmax, cap);
  }
}

// ---- striped Smith-Waterman: one 32-lane half per sequence, H and E in registers ---------------------------------
// SCAN (round 6, last session): the lazy-F correction as a prefix scan instead of the reference's loop.  What the loop computes
// (src/hhprefilter.cpp:176-203) is the vertical-gap value entering every stripe element from the elements above it,
//   Fin(k) = max(0, max over k' < k of ( Fend(k') - (k - 1 - k') * W * ge )),     Fend = the main pass's F behind its last row,
// then H[j] = max(H[j], Fin - j * ge) down the element's W rows: the loop finds it by shifting F one element per pass and stops at
// the first row where no element needs a correction - from there on nothing can change (the main pass's own F chain dominates:
// F <= H[j] - go gives F - ge <= H[j] - go <= the main pass's F of row j + 1 <= H[j + 1]; gap open >= gap extend, the host checks).
// The rows the loop never visits are rows the closed form does not change, so H - all that leaves a column - is the same; the
// column maximum is not touched by either (a corrected cell is below the cell its F came from); E is not updated in the loop (:175).
// The scan is a max-plus prefix over the 32 elements of a half (G = Fend + k W ge: four row_shr steps, one row_bcast:15 into the
// half's second row) - ~35 instructions a residue WHEN any element needs a correction (the loop's first test, kept as the one
// wave-uniform branch), whatever the sequence; the loop costs ~12 instructions per row it visits, 40-60 rows behind a high-scoring
// cell: the survivors of the gapless stage - homologs, what this kernel exists for - took 5.5 x the time of random sequences
// (20 000 x 300: 6.7 vs 1.2 ms; the reference's AVX2 loop 1.9 x), tools/bench_sw_homologs.py.
template <int W, bool SCAN>
__global__ void __launch_bounds__(512) hhv_pf_sw_kernel(PrefilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int WQ = (W + 3) / 4;
  uint32_t* sprof = reinterpret_cast<uint32_t*>(smem);
  fill_profile_lds<32, W>(sprof, a, 512);
  __syncthreads();
  const int k = threadIdx.x & 31;
  const int hsel = threadIdx.x & 32;  // which half of the wave
  const int cap = 255 - a.offset, go = a.gap_init, ge = a.gap_extend;
  const int64_t half_id = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 5), n_halves = (int64_t)gridDim.x * 16;
  const uint32_t* mine = sprof + k * WQ;
  const uint32_t* words = reinterpret_cast<const uint32_t*>(a.seqs);

  for (int64_t job = half_id; job < a.n_jobs; job += n_halves) {
    const int slot = a.order ? a.order[job] : (int)job;
    const int sid = a.subset ? a.subset[slot] : slot;
    const int64_t beg = a.offsets[sid];
    const int len = (int)(a.offsets[sid + 1] - beg);
    const int64_t w0 = beg >> 2;
    const int nw = (int)(((beg + len + 3) >> 2) - w0);
    const int head = (int)(beg & 3);
    int H[W], E[W];
#pragma unroll
    for (int t = 0; t < W; ++t) H[t] = E[t] = 0;
    int vmax = 0;
    // element k of the half keeps dword (chunk*32 + k) of the sequence, the next chunk is in flight; the residues are
    // taken a word (four of them) at a time so that the four profile reads are issued together, ahead of the arithmetic
    uint32_t chunk = 0, chunk_next = words[w0 + min(k, max(nw - 1, 0))];
    for (int wi = 0; wi < nw; ++wi) {
      if ((wi & 31) == 0) {
        chunk = chunk_next;
        chunk_next = words[w0 + min(wi + 32 + k, max(nw - 1, 0))];
      }
      const uint32_t word = (uint32_t)__shfl((int)chunk, wi & 31, 32);
      uint32_t pw[4][WQ];
#pragma unroll
      for (int b = 0; b < 4; ++b) read_cells<WQ>(mine + ((word >> (8 * b)) & 0xff) * (32 * WQ), pw[b]);
#pragma unroll
      for (int b = 0; b < 4; ++b) {
      const int rel = wi * 4 + b - head;  // residue index inside the sequence
      if (rel < 0 || rel >= len) continue;
      const uint32_t* p = pw[b];
      int F = 0;
      int h = half_shr1_zero(H[W - 1], k);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        h = med3i(h + sbyte(p[j >> 2], j & 3), cap);  // adds(H, profile), subs(H, bias)
        h = max(max(h, E[j]), F);
        vmax = max(vmax, h);
        const int old = H[j];
        H[j] = h;
        const int t = sat_sub(h, go);
        E[j] = max(sat_sub(E[j], ge), t);
        F = max(sat_sub(F, ge), t);
        h = old;
      }
      // lazy-F loop (:176-203); a half leaves it for good as soon as none of its 32 elements needs a correction.
      // The loop's first test - does any element need a correction at segment row 0 - fails for most residues, for both halves:
      // it is taken out of the loop as ONE wave-uniform test, so that the common case jumps over the whole (unrolled,
      // predicated) loop body with a single scalar branch instead of walking its W guarded blocks.
      const int Fend = F;
      F = half_shr1_zero(F, k);
      bool active = true;
      if (__ballot(sat_sub(F, sat_sub(H[0], go)) != 0) != 0) {
      if (SCAN) {
        // inclusive max-scan of G over the half: rows of 16 lanes (row_shr 1, 2, 4, 8; a lane without a source keeps its own
        // value), then the first row's last lane into the half's second row (row_bcast:15, rows 1 and 3 only)
        const int koff = k * (W * ge);
        int G = Fend + koff;
        G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x111, 0xF, 0xF, false));
        G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x112, 0xF, 0xF, false));
        G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x114, 0xF, 0xF, false));
        G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x118, 0xF, 0xF, false));
        G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x142, 0xA, 0xF, false));
        // exclusive: the element above; element 0 of a half has nothing above it
        const int up = __builtin_amdgcn_update_dpp(0, G, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
        int f = k == 0 ? 0 : max(up - (koff - W * ge), 0);
#pragma unroll
        for (int j = 0; j < W; ++j) {
          H[j] = max(H[j], f);
          f = sat_sub(f, ge);
        }
      } else
      for (;;) {
        bool done = false;
#pragma unroll
        for (int j = 0; j < W; ++j) {
          if (!done) {
            const bool need = sat_sub(F, sat_sub(H[j], go)) != 0;
            const bool any = ((__ballot(need && active) >> hsel) & 0xFFFFFFFFull) != 0;
            active = active && any;
            if (__ballot(active) == 0) {
              done = true;
            } else if (active) {
              H[j] = max(H[j], F);
              vmax = max(vmax, H[j]);
              F = sat_sub(F, ge);
            }
          }
        }
        if (done) break;
        const int Fs = half_shr1_zero(F, k);
        if (active) F = Fs;
      }
      }
      }  // b
    }    // wi
    for (int o = 16; o >= 1; o >>= 1) vmax = max(vmax, __shfl_xor(vmax, o, 32));
    if (k == 0) a.scores[slot] = vmax;
  }
}

// ---- generic kernel: H/E columns in LDS, profile striped like the reference in LDS or global --------------------
// STATE_GLOBAL: queries whose H/E columns (3 * 32 * W bytes per sequence slot) exceed LDS keep them in a.state_scratch; a
// thread only ever reads the bytes it wrote itself (stripe element k), so nothing but the address changes.
template <bool GAPPED, bool PROF_LDS, bool STATE_GLOBAL>
__global__ void __launch_bounds__(256) hhv_pf_generic_kernel(PrefilterArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int W = a.W;
  const int k = threadIdx.x & 31;     // stripe element
  const int half = threadIdx.x >> 5;  // 0..7: sequence slot inside the block
  unsigned char* sprof = smem;        // [220][W][32] when PROF_LDS
  unsigned char* state = STATE_GLOBAL ? a.state_scratch + ((size_t)blockIdx.x * 8 + half) * 3 * W * 32
                                      : smem + (PROF_LDS ? (size_t)220 * W * 32 : 0) + (size_t)half * 3 * W * 32;
  if (PROF_LDS) {
    // stripe the plain [220][Lq] profile exactly like Prefilter::stripe_query_profile (:386-425)
    for (int e = threadIdx.x; e < 220 * W * 32; e += 256) {
      const int kk = e & 31, j = (e >> 5) % W, x = (e >> 5) / W;
      const int p = kk * W + j;
      sprof[e] = (p >= a.Lq) ? (unsigned char)a.offset : a.profile[(size_t)x * a.Lq + p];
    }
    __syncthreads();
  }
  const unsigned char* prof = PROF_LDS ? sprof : a.striped;
  const int go = a.gap_init, ge = a.gap_extend, bias = a.offset;

  for (int64_t job = (int64_t)blockIdx.x * 8 + half; job < a.n_jobs; job += (int64_t)gridDim.x * 8) {
    const int slot = a.order ? a.order[job] : (int)job;
    const int sid = a.subset ? a.subset[slot] : slot;
    const unsigned char* seq = a.seqs + a.offsets[sid];
    const int len = (int)(a.offsets[sid + 1] - a.offsets[sid]);
    unsigned char* Ha = state;           // pvHStore
    unsigned char* Hb = state + W * 32;  // pvHLoad
    unsigned char* E = state + 2 * W * 32;
    for (int j = 0; j < W; ++j) {
      Ha[j * 32 + k] = 0;
      Hb[j * 32 + k] = 0;
      E[j * 32 + k] = 0;
    }
    int vmax = 0;
    for (int i = 0; i < len; ++i) {
      const unsigned char* P = prof + (size_t)seq[i] * W * 32;
      if (GAPPED) {
        int F = 0;
        int H = half_shr1_zero(Ha[(W - 1) * 32 + k], k);
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
        // lazy-F correction (:176-203).  gap open >= gap extend: as a prefix scan over the half's 32 stripe elements (see
        // hhv_pf_sw_kernel: the same H in every row), the rows walked once and only while the incoming F is above zero - for
        // other gap parameters the reference's loop, whose exit test then prunes the chain
        int j = 0;
        H = Ha[k];
        const int Fend = F;
        F = half_shr1_zero(F, k);
        bool need = max(0, F - max(0, H - go)) != 0;
        if (go >= ge) {
          if (((__ballot(need) >> (threadIdx.x & 32)) & 0xFFFFFFFFull) != 0) {   // (a half without a correction at row 0 has none at all)
            const int koff = k * (W * ge);
            int G = Fend + koff;
            G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x111, 0xF, 0xF, false));
            G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x112, 0xF, 0xF, false));
            G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x114, 0xF, 0xF, false));
            G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x118, 0xF, 0xF, false));
            G = max(G, __builtin_amdgcn_update_dpp(G, G, 0x142, 0xA, 0xF, false));
            const int up = __builtin_amdgcn_update_dpp(0, G, 0x138 /* wave_shr:1 */, 0xF, 0xF, true);
            int f = k == 0 ? 0 : max(up - (koff - W * ge), 0);
            for (j = 0; j < W && f > 0; ++j) {
              const int h0 = Ha[j * 32 + k];
              if (f > h0) Ha[j * 32 + k] = (unsigned char)f;
              f = ge > 0 ? max(0, f - ge) : f;
            }
          }
          need = false;
        }
        while (((__ballot(need) >> (threadIdx.x & 32)) & 0xFFFFFFFFull) != 0) {
          H = max(H, F);
          vmax = max(vmax, H);
          Ha[j * 32 + k] = (unsigned char)H;
          F = max(0, F - ge);
          ++j;
          if (j >= W) {
            j = 0;
            F = half_shr1_zero(F, k);
          }
          H = Ha[j * 32 + k];
          need = max(0, F - max(0, H - go)) != 0;
        }
      } else {
        // ungapped (:246-274): s_curr = Ha, s_prev = Hb
        int S = half_shr1_zero(Ha[(W - 1) * 32 + k], k);
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

// ---- launchers ----------------------------------------------------------------------------------------------------
template <typename K>
static int launch_one(K kernel, const PrefilterArgs& a, int n_blocks, int threads, size_t lds, void* stream) {
  (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kernel, dim3(n_blocks), dim3(threads), lds, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int)e;
}

size_t prefilter_fast_lds(bool gapped, int W) { return (size_t)(gapped ? 220 * 32 : 221 * 64) * ((W + 3) / 4) * 4; }

int launch_prefilter_fast(const PrefilterArgs& a, bool gapped, int W, int n_blocks, void* stream) {
  const size_t lds = prefilter_fast_lds(gapped, W);
  if (!gapped) {
    const bool slab = a.carry_in != nullptr || a.carry_out != nullptr || a.q_base != 0;
    // two cells per register: measured against the byte kernel in one session (profiles/r6_ab.txt, 1 M sequences): W = 2 (queries
    // of 65 .. 128 columns) 4.90 -> 4.24 ms; W = 4 7.5 = 7.5; W = 5 (300 columns) 9.6 -> 10.3 ms - the odd cell's halfword read,
    // its mask and the byte permute for the carry cost what the pairs save.  So: W = 2 only.  HHV_PF_PACKED=1 / 0 (measurement
    // aid): every W up to 5 / none.
    static const int packed_env = [] { const char* e = getenv("HHV_PF_PACKED"); return e ? atoi(e) : -1; }();
    const bool packed = packed_env < 0 ? W == 2 : packed_env != 0;
    if (packed && !slab && W <= 5) {
      const size_t lds_pk = (size_t)(PF_NULL + 1) * 64 * ((W / 2) * 4 + (W & 1) * 2);
      switch (W) {
        case 1: return launch_one(hhv_pf_ungapped_pk_kernel<1>, a, n_blocks, 1024, lds_pk, stream);
        case 2: return launch_one(hhv_pf_ungapped_pk_kernel<2>, a, n_blocks, 1024, lds_pk, stream);
        case 3: return launch_one(hhv_pf_ungapped_pk_kernel<3>, a, n_blocks, 1024, lds_pk, stream);
        case 4: return launch_one(hhv_pf_ungapped_pk_kernel<4>, a, n_blocks, 1024, lds_pk, stream);
        case 5: return launch_one(hhv_pf_ungapped_pk_kernel<5>, a, n_blocks, 1024, lds_pk, stream);
      }
    }
    switch (W) {
#define HHV_PF_CASE(w) \
  case w:              \
    return slab ? launch_one(hhv_pf_ungapped_kernel<w, true>, a, n_blocks, 1024, lds, stream) \
                : launch_one(hhv_pf_ungapped_kernel<w, false>, a, n_blocks, 1024, lds, stream);
      HHV_PF_CASE(1) HHV_PF_CASE(2) HHV_PF_CASE(3) HHV_PF_CASE(4) HHV_PF_CASE(5) HHV_PF_CASE(6) HHV_PF_CASE(7) HHV_PF_CASE(8)
#undef HHV_PF_CASE
    }
    return -(int)hipErrorInvalidValue;
  }
  // the scan form of the lazy-F correction needs gap open >= gap extend (always so in HH-suite: 20 / 4, src/hhdecl.cpp); HHV_PF_LAZY_LOOP=1:
  // the reference's loop (A/B, tests)
  static const bool lazy_loop = [] { const char* e = getenv("HHV_PF_LAZY_LOOP"); return e && atoi(e) == 1; }();
  const bool scan = !lazy_loop && a.gap_init >= a.gap_extend;
  switch (W) {
#define HHV_PF_CASE(w) \
  case w:              \
    return scan ? launch_one(hhv_pf_sw_kernel<w, true>, a, n_blocks, 512, lds, stream) \
                : launch_one(hhv_pf_sw_kernel<w, false>, a, n_blocks, 512, lds, stream);
    HHV_PF_CASE(1) HHV_PF_CASE(2) HHV_PF_CASE(3) HHV_PF_CASE(4) HHV_PF_CASE(5) HHV_PF_CASE(6) HHV_PF_CASE(7) HHV_PF_CASE(8)
    HHV_PF_CASE(9) HHV_PF_CASE(10) HHV_PF_CASE(11) HHV_PF_CASE(12) HHV_PF_CASE(13) HHV_PF_CASE(14) HHV_PF_CASE(15)
    HHV_PF_CASE(16) HHV_PF_CASE(17) HHV_PF_CASE(18) HHV_PF_CASE(19) HHV_PF_CASE(20)
#undef HHV_PF_CASE
  }
  return -(int)hipErrorInvalidValue;
}

int launch_prefilter_generic(const PrefilterArgs& a, bool gapped, bool prof_lds, int n_blocks, size_t lds, void* stream) {
  if (a.state_scratch)
    return gapped ? launch_one(hhv_pf_generic_kernel<true, false, true>, a, n_blocks, 256, 0, stream)
                  : launch_one(hhv_pf_generic_kernel<false, false, true>, a, n_blocks, 256, 0, stream);
  if (gapped)
    return prof_lds ? launch_one(hhv_pf_generic_kernel<true, true, false>, a, n_blocks, 256, lds, stream)
                    : launch_one(hhv_pf_generic_kernel<true, false, false>, a, n_blocks, 256, lds, stream);
  return prof_lds ? launch_one(hhv_pf_generic_kernel<false, true, false>, a, n_blocks, 256, lds, stream)
                  : launch_one(hhv_pf_generic_kernel<false, false, false>, a, n_blocks, 256, lds, stream);
}

}  // namespace hhv
