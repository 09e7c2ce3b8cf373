// hhv_stream_kernel.h -- the Viterbi DP kernel template (replaces Viterbi::Align, src/hhviterbialgorithm.cpp:29-497).
// Included by the instantiation units hhv_kernels.hip (W = 64 lanes per systolic array), hhv_kernels_w32.hip and
// hhv_kernels_w16.hip (short queries: two / four independent arrays per wavefront).  Per-lane arithmetic: viterbi_lane.h.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/hhviterbi_hip.h"  // HHV_SEGMENT_MIN_RECORDS: the planner's constant the ring geometry below depends on
#include "hhv_internal.h"
#include "viterbi_lane.h"

// which secondary-structure variants fetch the head of the next step early (PF below): all that have the registers for it
#ifndef HHV_SS_PF
#define HHV_SS_PF(R, LOCAL, BT, MULTI) (!((R) == 5 && (LOCAL)))  // (local mode, five rows: the per-row best leaves no room)
#endif

namespace hhv {

#if defined(HHV_EXP_TIMING)
// measurement build only: shader-clock totals of the three sections of a step, summed over the steps of wave 0 of block 0
// [0] steps, [1] top of the step + phase A, [2] phase B, [3] phase C + the end of the step
static __device__ unsigned long long hhv_dbg_clk[8];
// the stamp is one asm statement together with "+v" ties to values produced before / consumed after it, so that hipcc
// cannot move the arithmetic of the neighbouring phases across it
#define HHV_STAMP(t_, ...) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), __VA_ARGS__)
#endif

#if defined(HHV_EXP_WAVETIME)
// measurement build only (tools/wave_times.py): per workgroup the constant-rate 100 MHz clock at entry and exit, its range's
// record count and the hardware ids (HW_ID: simd / cu / sh / se, XCC_ID) - who finishes when, and where it ran
static __device__ unsigned long long hhv_dbg_wave[4 * 16384];
#endif

// Hand-off lane n <- lane n-1: pulled through the LDS crossbar (ds_bpermute_b32: no LDS memory, returns on lgkmcnt like a
// read) at the top of the step and delivered LATE - MM / DG / MI of the row above are first needed in phase C, GD / IM / DG as
// next step's diagonal - so the round trip (300+ clk under this kernel's LDS load) lies under phases A and B.  Rounds 1-2a
// used v_mov_b32_dpp wave_shr:1 / row_shr:1: with two waves per SIMD those moves cost ~10 % of the kernel (the launch with
// plain v_mov in their place - wrong results, timing only - 16.27 -> 14.67 ms, profiles/r2_ab_session2.txt ab9; replaying
// one step of the kernel's own ISA, tools/gen_step_replay_ubench.py, a DPP move holds up the other wave of the SIMD).
__device__ __forceinline__ void pull5(uint32_t addr, float& d0, float& d1, float& d2, float& d3, float& d4, float s0, float s1,
                                      float s2, float s3, float s4) {
  asm volatile("ds_bpermute_b32 %0, %5, %6\n\tds_bpermute_b32 %1, %5, %7\n\tds_bpermute_b32 %2, %5, %8\n\t"
               "ds_bpermute_b32 %3, %5, %9\n\tds_bpermute_b32 %4, %5, %10"
               : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4)
               : "v"(addr), "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(s4));
}
// Short-query arrays (W = 32 / 16 lanes, variants that read the record head at the top of the step): the DPP moves stay - the
// pulls, waited for with the head, measured 1.5 / 4 % slower there (profiles/r2_ab_session2.txt ab11).
// lane n <- lane n-1 of the same systolic array; the first lane of an array keeps `old`
//   W = 16: v_mov_b32_dpp row_shr:1  (a DPP row IS 16 lanes: lanes 0, 16, 32, 48 keep old)
//   W = 32: wave_shr:1, then lane 32 is put back to `old` (one v_cndmask with a loop-invariant lane mask)
template <int W>
__device__ __forceinline__ int dpp_shr1(int old, int src, bool first_of_array) {
  if (W == 16) return __builtin_amdgcn_update_dpp(old, src, 0x111, 0xF, 0xF, false);
  const int v = __builtin_amdgcn_update_dpp(old, src, 0x138, 0xF, 0xF, false);
  if (W == 32) return first_of_array ? old : v;
  return v;
}
template <int W>
__device__ __forceinline__ float dpp_shr1(float old, float src, bool first_of_array) {
  return __builtin_bit_cast(float, dpp_shr1<W>(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), first_of_array));
}

// ---- two-wave workgroups (hhv_pair_kernel): a query of two strips aligned in ONE launch ---------------------------------------
// The two strips of a query of 321 .. 640 rows used to be two launches, the bottom row of the first strip travelling to the
// second through HBM (20 bytes per stream record, a load per step in front of the hand-off: 6-10 % of those kernels, NOTES_r3 6,
// NOTES_r4 6).  Here the strips are the two wavefronts of ONE workgroup - a 128-lane systolic array: both walk the same
// sequence of stream segments (the first wave draws them from the queue and publishes the ids, the second follows), and
// the first wave's last lane hands its bottom row (and, at headers, the finalized best) to the second wave's first lane
// through a FIFO in LDS, slot = stream position mod PAIR_FIFO.  Flow control once per ring chunk, through two progress
// counters: the second wave runs 3 .. 9 chunks behind the first.
constexpr int PAIR_FIFO = 256;
struct PairLds {
  float4 carry[PAIR_FIFO][2];  // {MM, GD, IM, DG}, {MI, fs, fpos, -}
  int seg_id[16];              // segment ids in the order the first wave drew them
  int seg_count;               // ... how many so far
  int w0_done;                 // positions the first wave has written (INT_MAX when it is through)
  int w1_done;                 // positions the second wave has consumed
  int pad;
};
__device__ __forceinline__ PairLds* pair_lds() {
  __shared__ PairLds p;
  return &p;
}
// The control words are read and written through inline asm (a read and its wait in ONE statement): hipcc would put a
// vmcnt(0) in front of every LDS access it can see (the ring's LDS-DMA), and the audit (tools/audit_asm.py) allows no
// compiler-generated LDS read in the loops.
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ int lds_peek(uint32_t addr) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lds_poke(uint32_t addr, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// spin until the word at `addr` is >= need.  Bounded: a logic error (or a partner wave that is held up for tens of
// milliseconds) must not hang the device - but it must not pass for a result either: when the bound is hit the wave sets
// DEV_ERR_PAIR_TIMEOUT in the context's error word (host-mapped memory, a system-scope atomic OR like every reporter) and
// every call that waits for the stream answers HHV_E_DEVICE (hhv_api.cpp sync_check).  `dead` (wave
// uniform, kept by the caller) makes the later waits of a wave that has given up return at once: its results are garbage
// anyway, and a launch of thousands of chunks must not spend the bound thousands of times.
#if defined(HHV_EXP_PAIR_TIMEOUT)  // test build (tests/test_gpu_errors.py): the first wave never reports progress, short bound
constexpr int PAIR_WAIT_SPINS = 1 << 8;
#else
constexpr int PAIR_WAIT_SPINS = 1 << 17;
#endif
__device__ __forceinline__ void pair_wait(uint32_t addr, int need, uint32_t* err, int& dead) {
#if defined(HHV_EXP_PAIR_NOSYNC)  // measurement build, WRONG results: nobody waits for anybody
  return;
#endif
  if (dead) return;
  for (int guard = 0; guard < PAIR_WAIT_SPINS; ++guard) {
    if (lds_peek(addr) >= need) return;
    __builtin_amdgcn_s_sleep(4);
  }
  dead = 1;
  if (err) __hip_atomic_fetch_or(err, DEV_ERR_PAIR_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (an OR: concurrent reporters keep each other's bits)
}

// ---- work queue of the 64-lane variants (see the kernel: DQ) ------------------------------------
// A wave's stream is the concatenation of the segments it draws; positions count its records from 0.  All state is wave
// uniform and kept in SGPRs (every update goes through readfirstlane: these variants have no VGPR to spare).  A segment is
// drawn when the ring refill reaches the end of the current one - a round trip of an atomic and a scalar load per segment
// (>= 128 steps), during which the other wave of the SIMD has the issue slots to itself.
struct WorkQueue {
  int delta_lo, delta_hi;  // record = position + delta for positions < J
  int J;                   // end (position) of the current segment
  // the junction passed last, for the lanes' own record indices (backtrace entries): positions < Jlast have dprev, positions
  // >= Jlast delta_lo (mod 2^32: record indices are < 2^32).  One junction of history is enough: the refill runs at most 96
  // positions ahead of the first lane and segments have >= 128 records, so when it passes a junction the one before it is
  // behind the last lane (position s - 63).
  int Jlast, dprev;
  int tail;                // the refill has passed the terminal header: the rest is the stream's padding
  int end;                 // records of the wave's stream incl. the terminal header, M_OPEN until the queue is empty
  static constexpr int M_OPEN = 0x3FFFFFFF;

  static __device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ int64_t delta() const { return (int64_t)(((uint64_t)(uint32_t)delta_hi << 32) | (uint32_t)delta_lo); }
  __device__ __forceinline__ void set_delta(int64_t v) {
    delta_lo = uni((int)(uint32_t)v);
    delta_hi = uni((int)(uint32_t)((uint64_t)v >> 32));
  }
  // One ticket for the wave: EXEC is narrowed to lane 0 inside the statement (uniform control flow around it), and the
  // statement waits for the value itself - no register is in flight outside it.
  static __device__ __forceinline__ int draw(const uint32_t* queue) {
    uint64_t save;
    uint32_t zero = 0, one = 1, ticket;
    asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_waitcnt vmcnt(0)\n\t"
                 "s_mov_b64 exec, %1"
                 : "=&v"(ticket), "=&s"(save) : "v"(zero), "v"(one), "s"(queue) : "memory");
    return __builtin_amdgcn_readfirstlane((int)ticket);
  }
  // segment id = records [seg[2 id], seg[2 id + 1]) by one scalar load (the table is never written by a kernel)
  static __device__ __forceinline__ void segment(const int64_t* seg_first, int id, int64_t& first, int64_t& end) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i w;
    const uint32_t off = (uint32_t)id * 16u;
    asm volatile("s_load_dwordx4 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(seg_first), "s"(off) : "memory");
    first = (int64_t)(((uint64_t)(uint32_t)w.y << 32) | (uint32_t)w.x);
    end = (int64_t)(((uint64_t)(uint32_t)w.w << 32) | (uint32_t)w.z);
  }
  // first segment of an array: its own number (the ticket counter starts at the number of arrays - thousands of draws of the
  // same counter at the start of a launch would queue up behind each other); false: more arrays than segments, nothing to do
  __device__ __forceinline__ bool start(const int64_t* seg_first, int n_seg, int id) {
    Jlast = 0;
    dprev = 0;
    draws = 0;
    if (id >= n_seg) {
      delta_lo = delta_hi = J = 0;
      tail = 1;
      end = 0;
      return false;
    }
    int64_t f0, f1;
    segment(seg_first, id, f0, f1);
    set_delta(f0);
    J = uni((int)(f1 - f0));
    tail = 0;
    end = M_OPEN;
    return true;
  }
  // Refill of chunk cc = positions [C cc, C cc + C) of one systolic array (C = W / 2 records = 7 C float4, loaded by
  // ceil(7 C / 64) wave-wide 16-byte-per-lane loads straight into the array's part of the ring slot, `dst`).  When the chunk
  // reaches the end of the current segment the next one is drawn (queue empty: the terminal header, which ends the stream:
  // `end`); one junction at most lies inside a chunk (segments have >= 128 records; behind the terminal header the stream's
  // padding is read).  Everything that depends on the array is wave uniform here: no per-lane selects.
  // PM (pair mode, see PairLds): 1 = publish every id drawn, 2 = take the ids the first wave published instead of drawing
  int draws = 0;
  int dead = 0;  // pair kernels: this wave has given up waiting for its partner (pair_wait)
  template <int W, int PM = 0>
  __device__ __forceinline__ void refill(const float4* __restrict__ records, const int64_t* seg_first, int n_seg,
                                         const uint32_t* queue, int cc, float4* dst, int lane, PairLds* pair = nullptr,
                                         uint32_t* err = nullptr) {
    constexpr int C = W / 2, CF4 = C * 7, H = (CF4 + LANES - 1) / LANES;
    const int P0 = cc * C;
    const bool cross = !tail && P0 + C > J;
    int64_t next = 0;
    int nlen = 1;
    if (cross) {
      int id;
      if (PM == 2) {
        pair_wait(lds_addr_of(&pair->seg_count), draws + 1, err, dead);
        id = lds_peek(lds_addr_of(&pair->seg_id[draws & 15]));
        if (dead) id = n_seg;  // (given up: the id may never have been written - the stream ends here)
      } else {
        id = draw(queue);
        if (PM == 1) {  // (LDS executes a wave's operations in order: the id is written before the count)
          lds_poke(lds_addr_of(&pair->seg_id[draws & 15]), id);
          lds_poke(lds_addr_of(&pair->seg_count), draws + 1);
        }
      }
      draws = uni(draws + 1);
      int64_t f1;
      segment(seg_first, min(id, n_seg), next, f1);  // (entry n_seg = the terminal header)
      if (id < n_seg) {
        nlen = (int)(f1 - next);
      } else {
        end = uni(J + 1);
        tail = 1;
      }
    }
    const int64_t base = ((int64_t)P0 + delta()) * 7;
    const int64_t jump = cross ? ((next - J) - delta()) * 7 : 0;
    int ln = lane;
    asm volatile("" : "+v"(ln));  // (keeps hipcc from carrying the per-lane parts of the addresses through the step loop)
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const int x = h * LANES + ln;
      int64_t off = base + x;
      if (cross && P0 + ((x * 9363) >> 16) >= J) off += jump;  // x / 7 for x < 224
      if ((h + 1) * LANES <= CF4 || x < CF4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(records + off),
                                         (__attribute__((address_space(3))) void*)(dst + h * LANES), 16, 0, 0);
    }
    if (cross) {
      dprev = delta_lo;
      Jlast = J;
      set_delta(next - J);
      J = uni(J + nlen);
    }
  }
  // record index (mod 2^32) of position p, for p inside the lanes' window
  __device__ __forceinline__ uint32_t record_of(int p) const {
    return (uint32_t)p + (uint32_t)delta_lo + (p < Jlast ? (uint32_t)dprev - (uint32_t)delta_lo : 0u);
  }
};

// Ring refill.  Per refill every array of the wave gets its next chunk of C = W / 2 records; the 64 / W chunks together
// are always 32 records = 224 float4 = 3.5 wave-wide 16-byte-per-lane loads straight into LDS (one ring slot = the chunks
// of all arrays back to back).  W = 64: one stream, lane l of load k fetches float4 64 k + l of the chunk.  W < 64: float4
// x = 64 k + l of the slot belongs to array x / (7 C); its source is that array's stream (first record rb[array], a
// wave-uniform value per array selected per lane).  A chunk index beyond an array's last chunk is clamped (the slot is not
// read by that array any more; reading the stream's last chunk again keeps the access inside the allocation).
template <int W>
__device__ __forceinline__ void load_chunk(const float4* __restrict__ records, const int64_t (&rb)[LANES / W],
                                           const int (&nchunks)[LANES / W], int chunk, float4* ring, int lane) {
  constexpr int A = LANES / W, C = W / 2, CF4 = C * 7, SLOT_F4 = CHUNK_RECS * 7;  // 224 float4 per slot
  float4* l = ring + (chunk & (RING_CHUNKS - 1)) * SLOT_F4;
  if (A == 1) asm volatile("" : "+v"(lane));  // (64-lane variants: no VGPR pair for a per-lane address carried through the step loop)
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = k * LANES + lane;
    int64_t first = rb[0];
    int nch = nchunks[0], o = x;
    if (A > 1) {
      const int ar = x / CF4;
      o = x - ar * CF4;
#pragma unroll
      for (int j = 1; j < A; ++j) {
        first = (ar == j) ? rb[j] : first;
        nch = (ar == j) ? nchunks[j] : nch;
      }
    }
    const int cc = A > 1 ? max(0, min(chunk, nch - 1)) : chunk;  // W = 64: the caller stops at the last chunk
    const float4* g = records + (first * 7 + (int64_t)cc * CF4 + o);
    if (k < 3 || lane < 32)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(l + k * LANES), 16, 0, 0);
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

template <int R, bool QL_, bool SPLIT_ = false, bool LATE_ = true>
struct LdsColumn {
  static constexpr bool LATE = LATE_;  // the hand-off values are pulled through LDS and in flight until a wait (64-lane variants)
  static constexpr bool QL = QL_;
  // SPLIT_A (backtrace variants with the pipelined head): the phase-A query transitions are requested at the top of the
  // step without a wait; lane_column evaluates the MM candidates of all rows first (they do not need them) and calls
  // before_A2() in front of the GD / IM updates.
  static constexpr bool SPLIT_A = SPLIT_;
  static constexpr int NA = (R + 1) / 2;      // float4 reads covering the A block {m2i, i2i} x R (floats 0 .. 2R-1)
  static constexpr int C0 = (2 * R) / 4;      // first float4 of the C block {m2d, d2d} x R (floats 2R .. 4R-1)
  static constexpr int NC = R - C0;
  v4f v0, v1, v2, v3, v4, v5, v6;
  v4f qa0, qa1, qa2, qc0, qc1, qc2;
  // hand-off values pulled from lane - 1, in flight from the top of the step to the first lgkmcnt(0) behind it (GD / IM / DG
  // are pulled straight into st.dGD / dIM / dDG: their old values were consumed by lane_diag at the top of the step)
  float hMM, hMI, hfs;
  int hfpos;
  bool head_lane;  // first lane of an array: keeps the boundary instead of what it pulled
  // called behind a wait that covers the pulls (before_B / before_C in a column, header_tid on a header)
  template <class State>
  __device__ __forceinline__ Incoming resolve(const Incoming& bnd, State& st) {
    if (LATE) asm volatile("" : "+v"(hMM), "+v"(hMI), "+v"(st.dGD), "+v"(st.dIM), "+v"(st.dDG));
    Incoming r = bnd;
    r.MM = head_lane ? bnd.MM : hMM;
    r.GD = head_lane ? bnd.GD : st.dGD;
    r.IM = head_lane ? bnd.IM : st.dIM;
    r.DG = head_lane ? bnd.DG : st.dDG;
    r.MI = head_lane ? bnd.MI : hMI;
    return r;
  }
  __device__ __forceinline__ void resolve_best(const Incoming& bnd, Incoming& r) {
    r.fs = head_lane ? bnd.fs : hfs;
    r.fpos = head_lane ? bnd.fpos : hfpos;
  }
#if defined(HHV_EXP_TIMING)
  unsigned long long tA, tB;
#endif
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
  // SPLIT_A: the phase-A query transitions, issued without a wait (full EXEC, top of the step)
  __device__ __forceinline__ void qa_issue() {
    if (NA == 1) asm volatile("ds_read_b128 %0, %1" : "=&v"(qa0) : "v"(ql_addr) : "memory");
    if (NA == 2)
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(qa0), "=&v"(qa1) : "v"(ql_addr) : "memory");
    if (NA == 3)
      asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %3 offset:32"
                   : "=&v"(qa0), "=&v"(qa1), "=&v"(qa2) : "v"(ql_addr) : "memory");
  }
  // LDS returns a wave's reads in order: behind qa_issue() the step has requested the next head (2 reads) and the profile
  // (5 reads, begin_column), so "at most 7 outstanding" means the query transitions have landed
  __device__ __forceinline__ void before_A2() {
    if (!SPLIT_A) return;
    if (NA == 1) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(qa0));
    if (NA == 2) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(qa0), "+v"(qa1));
    if (NA == 3) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(qa0), "+v"(qa1), "+v"(qa2));
  }
  // software-pipelined head (score-only variants): the head of step s+1 is requested at the top of step s and waited for
  // at its end, so its LDS round trip lies under the step's arithmetic instead of in front of it
  static __device__ __forceinline__ void head_issue(uint32_t addr, v4f& n6, v4f& n5) {
    asm volatile("ds_read_b128 %0, %2 offset:96\n\tds_read_b128 %1, %2 offset:80" : "=&v"(n6), "=&v"(n5) : "v"(addr) : "memory");
  }
  static __device__ __forceinline__ void head_wait(v4f& n6, v4f& n5) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n6), "+v"(n5));
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
  // 64-lane variants: the finalized best of the template travels lane to lane through LDS memory - through the HEADER RECORD
  // itself (round 5; until then 8 bytes per lane behind the ring): dwords 2 and 3 of a header record are unused by the stream
  // format, every lane works on the record one step after its neighbour, so a lane leaves (fs, fpos) there at the end of its
  // header step and the next lane finds them next to the template index - ONE ds_read_b128 {index, L, fs, fpos} instead of three
  // reads, no cross-lane operation, no wait outside the header path, and no LDS of its own (which is what lets eight
  // wavefronts with LDS-parked query rows share a CU with the secondary-structure table: hhv_ss_kernel below).
  __device__ __forceinline__ int32_t header_tid_best() {
    v4f h;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(h) : "v"(rec_addr));
    const float t = h.x, p = h.w;
    hfs = h.z;
    hfpos = __builtin_bit_cast(int32_t, p);
    return __builtin_bit_cast(int32_t, t);
  }
  __device__ __forceinline__ void publish_best(float fs, int fpos) const {
    asm volatile("ds_write_b32 %0, %1 offset:8\n\tds_write_b32 %0, %2 offset:12" ::"v"(rec_addr), "v"(fs), "v"(fpos) : "memory");
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
#if defined(HHV_EXP_TIMING)
    HHV_STAMP(tA, "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4));
#endif
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
#if defined(HHV_EXP_TIMING)
  __device__ __forceinline__ void stamp_B(float& a, float& b) { HHV_STAMP(tB, "+v"(a), "+v"(b)); }
#endif
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

// Secondary-structure term of the 64-lane ...AndSS variants (hhv_ss_kernel; src/hhviterbialgorithm.cpp:194-213,278-280): the
// premultiplied table ssw * S33 / S73 / S37 lives in the workgroup's LDS; row[r] = LDS byte address of the table row of the
// lane's query row r, set once; per column the R values at row[r] + 4 * (column index of the record) are requested right behind
// the wait for the profile (they land under the emission arithmetic, like the phase-C query transitions of the QL sources) and
// waited for in front of phase C - reads and wait through inline asm like every other LDS access of the loop (a compiler-visible
// gather - round 4: global loads - drags a vmcnt(0) / lgkmcnt(0) of hipcc's choosing into every step).
template <int R>
struct SsLds {
  static constexpr bool ASYNC = true;
  uint32_t row[R];
  uint32_t col4;  // 4 x the column index (pred_index / dssp_index) of this step's record
  float v[R];
  __device__ __forceinline__ void issue() {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint32_t ad = row[r] + col4;
      asm volatile("ds_read_b32 %0, %1" : "=&v"(v[r]) : "v"(ad));
    }
  }
  __device__ __forceinline__ void wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(v[r]));  // (no consumer above the wait)
  }
  __device__ __forceinline__ float get(int r) const { return v[r]; }
};

// Occupancy: VALU issue needs >= 2 waves per SIMD to reach its rate on gfx950 (a lone wave issues one
// VOP2 every ~6.3 clk, two waves one every ~2.5 clk: tools/valu_ubench).  Up to R = 5 rows per lane the
// kernel is held to <= 256 VGPRs (2 waves/SIMD, no scratch in any variant).
// MULTI = the query needs more than one pass of 64*R rows (the carry hand-over code is compiled out otherwise).
// SS = secondary-structure term added to the emission score (the reference's ...AndSS builds).
// W = lanes per systolic array: 64, or - short queries, single pass - 32 / 16: the wave is 2 / 4 independent arrays, array
//     a = lane / W walking the stream range wave_rec[blockIdx * (64 / W) + a]; every array has its own ring section.
// LDS of one wavefront: [QL block][ring]
template <int R, bool BT, int W>
struct StreamSmem {
  // five-row backtrace variants: {m2i, i2i} and {m2d, d2d} of the lane's R query rows live in LDS (20 floats per lane; the
  // 20-dword stride is conflict-free for ds_read_b128) - the VGPRs they would occupy hold the compare results instead.
  static constexpr bool QL = BT && R == 5;  // (up to four rows per lane the registers hold the query's gap transitions as well)
  static constexpr int QL_F4 = QL ? LANES * 5 : 0;
  static constexpr int F4 = QL_F4 + RING_RECS * 7;
};

// The body of the kernel for ONE wavefront; `array0` = number of its first systolic array (the workgroup number for the
// one-wave kernel).  `smem` = its LDS (StreamSmem<R, BT, W>::F4 float4), owned by the kernel that calls it: the one-wave kernel
// has one, the two bodies of a pair kernel two disjoint ones, hhv_ss_kernel one per wavefront of its workgroup.
// `ss_tab` (64-lane SS variants): LDS byte address of the workgroup's copy of the premultiplied secondary-structure table.
// PM (pair mode): 0 = the wavefront is a workgroup of its own; 1 / 2 = first / second wavefront of a two-wave workgroup that
// aligns a query of two strips in ONE launch (hhv_pair_kernel below); 3 = a workgroup of its own that runs the FIRST strip of a
// one-launch-per-strip plan - known at compile time, so that its steps contain no code of the later strips (whose carry loads make
// hipcc wait vmcnt(0) in every step of a body that contains them, whichever role it plays at run time).
template <int R, bool LOCAL, bool BT, bool CELLOFF, bool MULTI, bool SS, int W, int PM>
__device__ __forceinline__ void stream_body(const StreamArgs& a, const int array0, float4* const smem, const uint32_t ss_tab = 0,
                                            PairLds* const pair_of_wave = nullptr) {
  const int lane = (int)(threadIdx.x & (LANES - 1));  // (the wavefront's lane: workgroups have one, two (pairs) or eight (hhv_ss_kernel) wavefronts)
  // PW0 / PW1: first / second wavefront of a pair.  PM 4 / 5 are the waves of a pair that is one link of a longer chain of
  // strips (queries of more than two strips run as a sequence of pair launches, round 4): 4 = first wave of a LATER pair - its
  // strip takes the bottom row of the strip above it from HBM like any later strip, and hands its own on through the FIFO;
  // 5 = second wave of a pair that is NOT the last - FIFO in, its own bottom row out to HBM for the next launch.
  constexpr bool PW0 = PM == 1 || PM == 4, PW1 = PM == 2 || PM == 5;
  constexpr bool PAIRED = PW0 || PW1;
  static_assert(!PAIRED || (MULTI && W == LANES && !CELLOFF && !(BT && R == 5)), "pair variants: two strips, 64 lanes, no cell-off, no LDS-parked query rows");
  static_assert(PM != 3 || MULTI, "first strip of a multi-strip plan");
  PairLds* const pair = PAIRED ? pair_of_wave : nullptr;  // (the FIFO and the control words of the two-wave workgroup / of this pair of the workgroup)
  const uint32_t pair_carry_addr = PAIRED ? (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)&pair->carry[0][0] : 0u;
  static_assert(W == 64 || W == 32 || W == 16, "lanes per array");
  static_assert(!MULTI || W == LANES, "short-query arrays are single pass");
  constexpr int A = LANES / W;   // arrays per wave
  constexpr int C = W / 2;       // records per ring chunk and array (the live window of W records spans <= 3 chunks)
  static_assert(RING_CHUNKS == 4 && CHUNK_RECS == 32 && A * C == CHUNK_RECS, "ring geometry");
  // SHARE: (MM(i-1,j-1) + q.M2M) and (MI(i-1,j-1) + q.M2M) are computed once and carried (viterbi_lane.h LaneState::aMM / aMI:
  // two additions per row and step less for ten registers)
  constexpr bool SHARE = true;
  constexpr bool QL = StreamSmem<R, BT, W>::QL;
  constexpr int QL_F4 = StreamSmem<R, BT, W>::QL_F4;
  float4* const ring = smem + QL_F4;
  const int g = lane & (W - 1);  // lane of the array
  const int arr = lane / W;
#if defined(HHV_EXP_WAVETIME)
  const unsigned long long wt_start = wall_clock64();
#endif
  // DQ: the 64-lane variants take their work from a queue of stream SEGMENTS (whole templates, >= 128 records
  // each, a.seg_first) instead of one fixed range per wave: a wave's stream is the concatenation of the segments it draws
  // (one atomic ticket each).  The position -> record mapping is needed by the ring refill and, in the backtrace / cell-off
  // variants, for the entry address (WorkQueue::record_of); the header of the next segment's first template finalizes the
  // previous template like any other header.  Why: waves run at different speeds (tools/wave_times.py: the two waves of a
  // SIMD finish 0.8 ms apart, XCDs differ by 3 %) and a fixed equal split ends with the slowest one - 4.6 % of the headline
  // launch, 15 % with mixed lengths (a fixed split cannot cut inside a 1000-column template either).
#if defined(HHV_NO_QUEUE)  // measurement build: a fixed range per array in every variant
  constexpr bool DQV = false;
#else
  constexpr bool DQV = true;
#endif
  WorkQueue wq[A] = {};
  // stream ranges: rb / M per lane (uniform within an array), and per array as wave-uniform values for the refill
  int64_t rb_a[A];
  int M_a[A], nch_a[A];
  int Mmax = 0;
  if (DQV) {
    bool any = false;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      any |= wq[j].start(a.seg_first, a.n_seg, array0 * A + j);
      rb_a[j] = wq[j].delta();
      M_a[j] = nch_a[j] = wq[j].end;
      Mmax = max(Mmax, M_a[j]);
    }
    if (!any) return;
  } else {
#pragma unroll
    for (int j = 0; j < A; ++j) {
      const int64_t b0 = a.wave_rec[array0 * A + j], e0 = a.wave_rec[array0 * A + j + 1];
      rb_a[j] = b0;
      M_a[j] = e0 > b0 ? (int)(e0 - b0) + 1 : 0;  // the range's records plus the next header (finalizes the last template)
      if (A == 1) {  // wave uniform, but loaded through the vector path: into SGPRs
        rb_a[j] = (int64_t)(((uint64_t)(uint32_t)WorkQueue::uni((int)(uint32_t)((uint64_t)b0 >> 32)) << 32) | (uint32_t)WorkQueue::uni((int)(uint32_t)b0));
        M_a[j] = WorkQueue::uni(M_a[j]);
      }
      nch_a[j] = (M_a[j] + C - 1) / C;
      Mmax = max(Mmax, M_a[j]);
    }
    if (Mmax == 0) return;
  }
  int64_t rb = rb_a[0];
  int M = M_a[0];
#pragma unroll
  for (int j = 1; j < A; ++j) {
    rb = (arr == j) ? rb_a[j] : rb;
    M = (arr == j) ? M_a[j] : M;
  }
  // work queue: the refill of chunk cc for every array that still has records there; then the arrays' stream ends as the
  // lanes see them (M per lane, Mmax for the loop) and - backtrace / cell-off variants of the short-query arrays - the lane's
  // copy of its array's position -> record mapping (the 64-lane variants read the SGPRs)
  uint32_t lq_delta = 0, lq_prev = 0;
  int lq_J = 0;
  auto dq_refill = [&](const int cc) __attribute__((always_inline)) {
    constexpr int SLOT_F4 = CHUNK_RECS * 7;
    float4* const slot = ring + (cc & (RING_CHUNKS - 1)) * SLOT_F4;
#pragma unroll
    for (int j = 0; j < A; ++j) {
      if (cc * C < wq[j].end) wq[j].template refill<W, (PW0 ? 1 : PW1 ? 2 : 0)>((const float4*)a.records, a.seg_first, a.n_seg, a.queue, cc, slot + j * (C * 7), lane, pair, a.err);
    }
    M = wq[0].end;
    Mmax = wq[0].end;
#pragma unroll
    for (int j = 1; j < A; ++j) {
      M = (arr == j) ? wq[j].end : M;
      Mmax = max(Mmax, wq[j].end);
    }
    if (A > 1 && (BT || CELLOFF)) {
      lq_delta = (uint32_t)wq[0].delta_lo, lq_prev = (uint32_t)wq[0].dprev, lq_J = wq[0].Jlast;
#pragma unroll
      for (int j = 1; j < A; ++j) {
        lq_delta = (arr == j) ? (uint32_t)wq[j].delta_lo : lq_delta;
        lq_prev = (arr == j) ? (uint32_t)wq[j].dprev : lq_prev;
        lq_J = (arr == j) ? wq[j].Jlast : lq_J;
      }
    }
  };
  auto record_of = [&](const int p) __attribute__((always_inline)) -> uint32_t {
    if (A == 1) return wq[0].record_of(p);
    return (uint32_t)p + lq_delta + (p < lq_J ? lq_prev - lq_delta : 0u);
  };

  Params P;
  P.negq = a.negq;
  P.negt = a.negt;
  P.shift = a.shift;
  P.Lq = a.Lq;
  // log2f4's constants as SGPR operands instead of 32-bit literals (viterbi_lane.h: Log2Consts; expor is the v_alignbit
  // operand 0x4B000000 >> 9); the asm keeps hipcc from folding them back into the instructions
  asm volatile("s_mov_b32 %0, 0xbddba835\n\ts_mov_b32 %1, 0x3f3030c0\n\ts_mov_b32 %2, 0xbfe0d411\n\ts_mov_b32 %3, 0x402786ee\n\t"
               "s_mov_b32 %4, 0x4b00007f\n\ts_mov_b32 %5, 0x00258000\n\ts_mov_b32 %6, 0x007fffff"
               : "=s"(P.lg.c4), "=s"(P.lg.c3), "=s"(P.lg.c2), "=s"(P.lg.c1), "=s"(P.lg.ebias), "=s"(P.lg.expor), "=s"(P.lg.mant));
  const int i0 = a.row_base + g * R + 1;
  // the lane that emits results: owner of row Lq in the last pass, the array's last lane otherwise
  const int g_last = (!MULTI || a.pass_last) ? (a.Lq - a.row_base - 1) / R : W - 1;
  const int r_last = (a.Lq - a.row_base - 1) % R;
#if defined(HHV_EXP_MULTI_NOCARRY)  // measurement build, WRONG results: the multi-pass bodies without their carry traffic
  const bool first = true;
  const bool carry_out = false;
#else
  // (the waves of a pair know their role at compile time: no code of the other role - its global loads made hipcc put a
  // vmcnt(0) into every step of BOTH roles, profiles/r4_ab.txt ab-r4-5)
  const bool first = (PM == 1 || PM == 3) ? true : (PM == 2 || PM == 4 || PM == 5) ? false : (!MULTI || a.pass_first != 0);
  const bool carry_out = PM == 5 ? true : PAIRED ? false : (MULTI && a.pass_last == 0);
#endif

  const float4* const records = (const float4*)a.records;
  int nchunks_max = (Mmax + C - 1) / C;
  if (DQV) {
    dq_refill(0);
    dq_refill(1);  // (a stream of less than two chunks is at its end already: M, Mmax)
    nchunks_max = (Mmax + C - 1) / C;
  } else {
    load_chunk<W>(records, rb_a, nch_a, 0, ring, lane);
    load_chunk<W>(records, rb_a, nch_a, 1, ring, lane);  // the stream is padded: over-reading past M is harmless
  }

  QRows<R> q;
  q.load(a.qpack + (size_t)g * R * REC_DW);
  // PF: the head of the NEXT step (7 transitions + meta) is fetched while the current one is computed (8 more VGPRs).
  // The ring bookkeeping then runs one step early: chunk c must have landed before step C c - 1 requests its first record.
  // All 64-lane variants except the cell-off ones (no VGPRs to spare there; the secondary-structure variants joined in round 5,
  // when their table values stopped being compiler-scheduled global loads); in the backtrace
  // variants the phase-A query-transition reads are deferred as well (SPLIT_A).  Measured in one session each
  // (tools/gpu_ab.sh): score-only -2 %, multi-pass -0.8 %, backtrace -0.5 %.
  // (not the local five-row single-pass variants: 256 VGPRs do not hold the prefetched head next to the per-row best)
  // SSL: the secondary-structure values come out of the workgroup's LDS table (SsLds; 64-lane arrays = hhv_ss_kernel); the
  // short-query arrays gather them from global memory as before
  constexpr bool SSL = SS && W == LANES;
  constexpr bool PF = !CELLOFF && (!SS || (SSL && HHV_SS_PF(R, LOCAL, BT, MULTI))) && W == LANES && !(LOCAL && R == 5 && !MULTI);
  LdsColumn<R, QL, (PF && QL), (W == LANES)> col;
  const uint32_t smem_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
  const uint32_t ring_addr = smem_addr + QL_F4 * 16 + arr * (C * REC_DW * 4);
  col.ql_addr = smem_addr + lane * 80;
  col.head_lane = W == LANES && g == 0;  // (the DPP moves of the short-query arrays deliver the boundary themselves)
  col.hMM = col.hMI = col.hfs = NEG_MAX;
  col.hfpos = 0;
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
  int ss_qoff[SSL ? 1 : R];
  SsLds<R> ssl;
  if (SS && !SSL) {
#pragma unroll
    for (int r = 0; r < R; ++r) ss_qoff[r] = a.ss_q_off[i0 - 1 + r];
  }
  if (SSL) {
#pragma unroll
    for (int r = 0; r < R; ++r) ssl.row[r] = ss_tab + 4u * (uint32_t)a.ss_q_off[i0 - 1 + r];
  }
  // (the column index of a record as a byte offset into a table row: (meta >> (shift - 2)) & (mask << 2), both wave uniform)
  const int ss_sh4 = a.ss_t_shift - 2, ss_mask4 = a.ss_t_mask << 2;

  // chunks 0 and 1 have landed, the lane's own ds_writes above are done (LDS executes a wave's operations in order)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  constexpr int LEAD = PF ? 1 : 0;
  //  WorkQueue::record_of keeps ONE junction of history: when the refill (at most 2 C + LEAD positions ahead of the
  // first lane) passes a junction, the one before it must be behind the array's last lane (W - 1 positions back), and a ring
  // chunk may hold one junction only.  The planner's constant, the prefetch lead and the ring geometry are tied together here.
  static_assert(HHV_SEGMENT_MIN_RECORDS >= 2 * C + W - 2 + LEAD, "segments too short for one junction of history");
  static_assert(HHV_SEGMENT_MIN_RECORDS >= CHUNK_RECS, "a ring chunk would hold two junctions");
  // the ring arithmetic's constants as SGPR operands as well (three more literal-carrying instructions per step otherwise)
  uint32_t k_ring_mask, k_rec_bytes, k_jmask;
  asm volatile("s_mov_b32 %0, %3\n\ts_mov_b32 %1, %4\n\ts_mov_b32 %2, %5"
               : "=s"(k_ring_mask), "=s"(k_rec_bytes), "=s"(k_jmask) : "i"(RING_RECS - 1), "i"(REC_DW * 4), "i"(META_JMASK));
  auto record_addr = [&](int step) -> uint32_t {
    const int rr = step - g;
    if (W == LANES) return ring_addr + ((uint32_t)rr & k_ring_mask) * k_rec_bytes;
    // slot (r / C) & 3 of the ring, record r % C of this array's section of the slot
    const uint32_t t = (uint32_t)rr & (4 * C - 1);
    return ring_addr + (t / C) * (CHUNK_RECS * REC_DW * 4) + (t & (C - 1)) * (REC_DW * 4);
  };
  // PF: two operand sources used in turn - the step that computes with `cur` fetches the head of the next step into
  // `nxt` (no copy between the steps: the step loop is unrolled by two).
  decltype(col) col2 = col;
  if (PF) {
    col.rec_addr = record_addr(0);
    decltype(col)::head_issue(col.rec_addr, col.v6, col.v5);
    decltype(col)::head_wait(col.v6, col.v5);
  }

  // Later passes of a long query: lane 0 takes the bottom row the previous pass left for its record (a.carry).  The row of
  // step s + 1 is requested during step s - a global round trip per step in front of the hand-off would stall the wave
  // (and with two waves per SIMD, most of the SIMD) for its whole latency.
  float4 ncar = make_float4(0.f, 0.f, 0.f, 0.f);
  float nmi = 0.f;
  // (Round 4 tried the rows in BLOCKS instead: once per ring chunk lanes 0 .. 31 load the rows of the next chunk's positions -
  // one coalesced load a whole chunk ahead, retired by the chunk boundary's own vmcnt(0) - and lane 0 takes its row out of the
  // lane that holds it with five v_readlane per step: no global load, no address arithmetic, no wait in the step - and 2-4 %
  // SLOWER on every multi-pass query (Lq 431: 12.17 -> 12.65 ms; profiles/r4_ab.txt ab-r4-3): a VALU write to an SGPR costs
  // more than the load it replaces, like the v_cmp_e64 of round 3.)
  // PM == 2: the row comes out of the pair's FIFO instead (slot = position mod PAIR_FIFO): requested one step ahead like the
  // global row, by an inline-asm read that is waited for by the step's own lgkmcnt(0) waits (every lane reads the slot:
  // a broadcast; lane 0 uses it).  pc0 = {MM, GD, IM, DG}, pc1 = {MI, fs, fpos, -} of the NEXT step's position.
  v4f pc0 = {0.f, 0.f, 0.f, 0.f}, pc1 = {0.f, 0.f, 0.f, 0.f};
  auto pair_carry_issue = [&](const int pos) __attribute__((always_inline)) {
#if defined(HHV_EXP_PAIR_NOFIFO)  // measurement build, WRONG results: no carry traffic through LDS
    return;
#endif
    const uint32_t addr = pair_carry_addr + ((uint32_t)pos & (uint32_t)(PAIR_FIFO - 1)) * 32u;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(pc0), "=&v"(pc1) : "v"(addr) : "memory");
  };
  if (PW1) {
    pair_wait(lds_addr_of(&pair->w0_done), C + 1, a.err, wq[0].dead);  // the first chunk's positions (and the one read ahead) have been written
    pair_carry_issue(0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(pc0), "+v"(pc1));
  }
  if (MULTI && !first && !PW1) {
    if (lane == 0 && M > 0) {
      ncar = a.carry[rb];  // (work queue: rb = the first record of the wave's first segment)
      nmi = a.carry_mi[rb];
    }
  }
#if defined(HHV_EXP_TIMING)
  unsigned long long dbg0 = 0, dbg1 = 0, dbg2 = 0, dbg3 = 0;
#endif
  // every lane pulls from the lane below it (the first lane of an array pulls whatever and keeps the boundary instead)
  const uint32_t pull_addr = (uint32_t)((lane + LANES - 1) & (LANES - 1)) * 4u;

  auto step = [&](const int s, decltype(col)& cur, decltype(col)& nxt) __attribute__((always_inline)) {
    const int r = s - g;
    const bool active = (uint32_t)r < (uint32_t)M;  // 0 <= r < M in one compare
#if defined(HHV_EXP_TIMING)
    unsigned long long t0;
    HHV_STAMP(t0, "+v"(cur.v6), "+v"(cur.v5));
    cur.tA = t0;
    cur.tB = t0;
#endif

    // (PF: the address of this step's record was formed one step ago, for the head prefetch, and kept in the operand source)
    if (!PF) cur.rec_addr = record_addr(s);

    // hand-off from lane g-1 (full EXEC here).  The row-0 sums that read LAST step's hand-off first; then the pulls: MM and MI
    // into the operand source, GD / IM / DG straight into st.dGD / dIM / dDG - nothing touches the five registers until
    // LdsColumn::resolve(), behind a wait (the pulls are issued in front of this step's other reads: lgkmcnt counts stay valid)
    DiagSums ds;
    if (W == LANES) {
      ds = lane_diag(st, q);
      asm volatile("" : "+v"(ds.t2), "+v"(ds.x3), "+v"(ds.x4));  // (hipcc would sink the three adds below the pulls and copy)
    }
    int32_t meta;
    if (W < LANES) {
      // short-query arrays: DPP moves.  A move never writes the first lane of an array, which therefore keeps the boundary
      // (-FLT_MAX, position 0) it was initialised with for the whole kernel: only MM of row 0, which changes per column,
      // goes through the `old` operand
      cur.head();
      meta = cur.meta();
      ds = lane_diag(st, q);
      asm volatile("" : "+v"(ds.t2), "+v"(ds.x3), "+v"(ds.x4));
      const Incoming b0 = boundary_incoming(meta, meta & (int)k_jmask, P);
      const bool first_lane = g == 0;
      cur.hMM = dpp_shr1<W>(b0.MM, st.MM[R - 1], first_lane);
      st.dGD = dpp_shr1<W>(st.dGD, st.GD[R - 1], first_lane);
      st.dIM = dpp_shr1<W>(st.dIM, st.IM[R - 1], first_lane);
      st.dDG = dpp_shr1<W>(st.dDG, st.DG[R - 1], first_lane);
      cur.hMI = dpp_shr1<W>(cur.hMI, st.MI[R - 1], first_lane);
      if (__builtin_amdgcn_ballot_w64(meta < 0) != 0) {
        cur.hfs = dpp_shr1<W>(cur.hfs, st.fs, first_lane);
        cur.hfpos = dpp_shr1<W>(cur.hfpos, st.fpos, first_lane);
      }
    } else if (PF) {
      pull5(pull_addr, cur.hMM, st.dGD, st.dIM, st.dDG, cur.hMI, st.MM[R - 1], st.GD[R - 1], st.IM[R - 1], st.DG[R - 1], st.MI[R - 1]);
      meta = cur.meta();  // (the head of this step landed at the end of the previous one)
      if (QL) cur.qa_issue();
      nxt.rec_addr = record_addr(s + 1);
      decltype(col)::head_issue(nxt.rec_addr, nxt.v6, nxt.v5);
    } else {
      // 64-lane variants that read the head at the top of the step: the pulls in front of it, its wait covers them
      pull5(pull_addr, cur.hMM, st.dGD, st.dIM, st.dDG, cur.hMI, st.MM[R - 1], st.GD[R - 1], st.IM[R - 1], st.DG[R - 1], st.MI[R - 1]);
      cur.head();
      meta = cur.meta();
    }
    // what the first lane of an array takes instead: the DP boundary row 0, or - in later passes of a long query - the
    // bottom row the previous pass left for this record (and its running best)
    const int jcol = meta & (int)k_jmask;  // column index of a column record
    Incoming in = boundary_incoming(meta, jcol, P);
    if (PW1) {
      // (pc0 / pc1 were requested a step ago and have landed: the end of every step waits lgkmcnt(0))
      if (lane == 0 && active) {
        in.MM = pc0.x;
        in.GD = pc0.y;
        in.IM = pc0.z;
        in.DG = pc0.w;
        in.MI = pc1.x;
        if (meta < 0 && st.tid >= 0) {
          in.fs = pc1.y;
          const float fp = pc1.z;
          in.fpos = __builtin_bit_cast(int, fp);
        }
      }
      pair_carry_issue(s + 1);
    } else if (MULTI && !first) {
      if (lane == 0 && active) {
        in.MM = ncar.x;
        in.GD = ncar.y;
        in.IM = ncar.z;
        in.DG = ncar.w;
        in.MI = nmi;
        if (meta < 0 && st.tid >= 0) {
          const DevResult pr = a.results[st.tid & TID_MASK];
          in.fs = pr.score;
          in.fpos = (pr.i2 << 16) | pr.j2;
        }
      }
      // The carry row of the next step is requested now, behind the copies that consumed this step's row - by EVERY lane, for
      // its own next position (clamped into the stream: 64 neighbouring rows, one coalesced load; lane 0's is the one that is
      // used).  Round 4: requested by lane 0 alone, under a divergent branch, the loaded registers met their old values in a
      // phi behind the branch and hipcc waited vmcnt(0) right there - a memory round trip in every step, +24-26 % per step
      // (profiles/r4_ab.txt ab-r4-4).  Without the branch the first use is the top of the next step.
      // (Letting the other lanes look AHEAD instead - positions s + 2 .. s + 16, so that lane 0's row is in the cache when its turn
      // comes - changed nothing: the trip to HBM was not what was left.  profiles/r4_ab.txt ab-r4-7.)
      {
        const int pn = min(max(r + 1, 0), M - 1);
        const size_t rn = DQV ? (size_t)record_of(pn) : (size_t)(rb + pn);
        ncar = a.carry[rn];
        nmi = a.carry_mi[rn];
      }
    }

    if (active) {
      if (meta < 0) {
        TemplateResult res;
        // (the wait inside covers the pulls of this step as well)
        const int tid0 = W == LANES ? cur.header_tid_best() : cur.header_tid();
        const int new_tid = tid0 | ((meta & META_NOLASTCOL) ? TID_NOLASTCOL : 0);
        Incoming inh = cur.resolve(in, st);
        cur.resolve_best(in, inh);
        const bool emit = lane_header<R, LOCAL, SHARE>(st, q, inh, i0, new_tid, P, g == g_last, res);
        if (W == LANES) cur.publish_best(st.fs, st.fpos);
        if (emit && !PW0) {  // (first wave of a pair: the best travels on through the FIFO)
          DevResult o;
          o.score = res.score;
          o.i2 = res.i2;
          o.j2 = res.j2;
          o.index = res.tid;
          a.results[res.tid] = o;
        }
      } else {
        const int j = jcol;
        uint64_t cell = 0;
        uint64_t* bte = nullptr;
        // entry of (record rb + r, lane g): row rb + r + g = rb + s, the same row for all lanes of the array (bt_entry)
        if (BT || CELLOFF) {
          // (work queue: the lane's record sits wherever its segment lies; (record + g) is the row, uniform between junctions)
          const size_t row = DQV ? (size_t)(uint32_t)(record_of(r) + (uint32_t)g) : (size_t)(rb + s);
          bte = a.bt + ((MULTI ? (size_t)a.bt_plane * a.bt_pass_stride : 0) + row * W + g);
        }
        if (CELLOFF) cell = *bte;
        float ssv[SSL ? 1 : R];
        if (SS && !SSL) {
          const int tidx = (meta >> a.ss_t_shift) & a.ss_t_mask;
#pragma unroll
          for (int r = 0; r < R; ++r) ssv[r] = a.ss_table[ss_qoff[r] + tidx];
        }
        if (SSL) ssl.col4 = (uint32_t)((meta >> ss_sh4) & ss_mask4);
        PtrSs ssp{ssv};
        // single pass: the boundary value of the first lane is formed inside the column's block, so that it need not be held
        // through phases A and B (it is read in phase C)
        const Incoming inc = MULTI ? in : boundary_incoming(meta, jcol, P);
        uint64_t bytes;
        if (SSL) bytes = lane_column<R, LOCAL, BT, CELLOFF, SHARE, SS, bt_mm_mode(W, R, LOCAL, CELLOFF, SS), bt_pair_mode(W, CELLOFF)>(st, q, inc, ds, cur, j, i0, r_last, P, cell, ssl);
        else bytes = lane_column<R, LOCAL, BT, CELLOFF, SHARE, SS, bt_mm_mode(W, R, LOCAL, CELLOFF, SS), bt_pair_mode(W, CELLOFF)>(st, q, inc, ds, cur, j, i0, r_last, P, cell, ssp);
        if (BT) *bte = bytes;
      }
      if (PW0) {
#if !defined(HHV_EXP_PAIR_NOFIFO)
        if (lane == LANES - 1) {
          const uint32_t addr = pair_carry_addr + ((uint32_t)r & (uint32_t)(PAIR_FIFO - 1)) * 32u;
          const v4f o0 = {st.MM[R - 1], st.GD[R - 1], st.IM[R - 1], st.DG[R - 1]};
          const v4f o1 = {st.MI[R - 1], st.fs, __builtin_bit_cast(float, st.fpos), 0.0f};
          asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16" ::"v"(addr), "v"(o0), "v"(o1) : "memory");
        }
#endif
      } else if (carry_out) {
        if (lane == LANES - 1) {
          const size_t rc = DQV ? (size_t)record_of(r) : (size_t)(rb + r);
          a.carry[rc] = make_float4(st.MM[R - 1], st.GD[R - 1], st.IM[R - 1], st.DG[R - 1]);
          a.carry_mi[rc] = st.MI[R - 1];
        }
      }
    }
    if (PF) decltype(col)::head_wait(nxt.v6, nxt.v5);  // full EXEC again: the head of step s+1 is in nxt.v6 / v5 from here on
    // Multi-strip bodies, unrolled: hipcc lays the column / header blocks (which hold the waits for this step's pulls) out of
    // line, hoists the next step's row-0 sums above the wait above, and re-uses the register of the pulled MI for the carry-out
    // address.  All of that is behind a wait on every path that has an active lane - but the five pull targets are tied to
    // this point anyway, so that the order is in the code (and tools/audit_asm.py, which reads the ISA top to bottom, can see it).
    if (PF && MULTI && W == LANES && R < 5) asm volatile("" : "+v"(st.dGD), "+v"(st.dIM), "+v"(st.dDG), "+v"(cur.hMM), "+v"(cur.hMI));
    // (pair: the FIFO slot requested at the top of this step has landed too - LDS returns a wave's reads in order; the tie
    // keeps its eight registers the slot's until here, whatever the step uses of them)
    if (PW1) asm volatile("" : "+v"(pc0), "+v"(pc1));
#if defined(HHV_EXP_TIMING)
    {
      unsigned long long tE;
      HHV_STAMP(tE, "+v"(nxt.v6), "+v"(nxt.v5));
      dbg0 += 1;
      dbg1 += cur.tA - t0;
      dbg2 += cur.tB - cur.tA;
      dbg3 += tE - cur.tB;
    }
#endif
  };

  // Two nested loops: the outer one walks the ring chunks (one refill each), the inner one the C steps of a chunk, so
  // that the refill test is not part of a step.
  int s_end = Mmax + W - 1;
  for (int c = 0; c * C - LEAD < s_end; ++c) {
    if (c > 0) {
      // chunk c was issued C steps ago: make sure it has landed, then refill the slot that held chunk c-3 (its last
      // reader, lane W-1, finished at step C(c-2)+2(W-1) < C c - LEAD) with chunk c+1.  The live window [s-W+1, s] spans
      // chunks c-2..c, so the ring holds 4 chunks = 2W records per array.
      // (The same wait retires the backtrace stores of the last C steps - the only place they are waited for.)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (PW0) {
        // positions below c C - LEAD - (W - 1) are in the FIFO (their ds_writes are complete: LDS executes a wave's operations
        // in order and the steps' waits have passed them); the coming chunk writes positions up to pmax, which must not lap the
        // second wave
        #if !defined(HHV_EXP_PAIR_TIMEOUT)
        if (c * C - LEAD - (W - 1) > 0) lds_poke(lds_addr_of(&pair->w0_done), c * C - LEAD - (W - 1));
#endif
        const int pmax = (c + 1) * C - LEAD - W;
        if (pmax - (PAIR_FIFO - 1) > 0) pair_wait(lds_addr_of(&pair->w1_done), pmax - (PAIR_FIFO - 1), a.err, wq[0].dead);
      }
      if (PW1) {
        lds_poke(lds_addr_of(&pair->w1_done), c * C - LEAD);
        pair_wait(lds_addr_of(&pair->w0_done), (c + 1) * C - LEAD + 1, a.err, wq[0].dead);
      }
      if (DQV) {
        const int m_before = Mmax;
        if (c + 1 < nchunks_max) dq_refill(c + 1);
        if (Mmax != m_before) {  // the queue is empty: the last array's stream ends behind its current segment's terminal header
          s_end = Mmax + W - 1;
          nchunks_max = (Mmax + C - 1) / C;
          if (c * C - LEAD >= s_end) break;
        }
      } else {
        if (c + 1 < nchunks_max) load_chunk<W>(records, rb_a, nch_a, c + 1, ring, lane);
      }
    }
    const int s_lo = c > 0 ? c * C - LEAD : 0, s_hi = min((c + 1) * C - LEAD, s_end);
    if (PF && (!MULTI || R < 5)) {
      // unrolled by two: the heads alternate between col and col2, no copy between the steps.  (Round 3 had measured the
      // multi-strip variants 1 % slower unrolled - with hipcc's vmcnt(0) behind the carry loads in every step; without it the
      // five copies per step count: Lq 512 -2.4 %, Lq 431 -2.3 %, Lq 1000 -1.6 %, with backtrace -0.5 .. -1 %, ab-r4-8.
      // Five-row strips of multi-strip queries have no registers for the second set of heads: they spill, and keep the copies.)
      int s = s_lo;
      for (; s + 1 < s_hi; s += 2) {
        step(s, col, col2);
        step(s + 1, col2, col);
      }
      if (s < s_hi) {  // odd number of steps (the first chunk, the last one): once per chunk
        step(s, col, col2);
        col.v6 = col2.v6;
        col.v5 = col2.v5;
        col.rec_addr = col2.rec_addr;
      }
    } else if (PF) {
      for (int s = s_lo; s < s_hi; ++s) {
        step(s, col, col2);
        col.v6 = col2.v6;
        col.v5 = col2.v5;
        col.rec_addr = col2.rec_addr;
      }
    } else {
      for (int s = s_lo; s < s_hi; ++s) step(s, col, col);
    }
  }
#if !defined(HHV_EXP_PAIR_TIMEOUT)
  if (PW0) lds_poke(lds_addr_of(&pair->w0_done), 0x7FFFFFFF);  // through: the second wave never waits again
#endif
  if (PW1) lds_poke(lds_addr_of(&pair->w1_done), 0x7FFFFFFF);
#if defined(HHV_EXP_WAVETIME)
  if (lane == 0 && array0 < 16384) {
    hhv_dbg_wave[4 * array0 + 0] = wt_start;
    hhv_dbg_wave[4 * array0 + 1] = wall_clock64();
    hhv_dbg_wave[4 * array0 + 2] = (unsigned long long)Mmax;
    hhv_dbg_wave[4 * array0 + 3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);
  }
#endif
#if defined(HHV_EXP_TIMING)
  if (array0 == 0 && lane == 0) {
    hhv_dbg_clk[0] = dbg0;
    hhv_dbg_clk[1] = dbg1;
    hhv_dbg_clk[2] = dbg2;
    hhv_dbg_clk[3] = dbg3;
  }
#endif
}

// FIRSTP: the launch of the FIRST strip of a multi-strip query (MULTI only)
template <int R, bool LOCAL, bool BT, bool CELLOFF, bool MULTI, bool SS, int W, bool FIRSTP = false>
__global__ void __launch_bounds__(LANES, 2) hhv_stream_kernel(StreamArgs a) {
  static_assert(!(SS && W == LANES), "the 64-lane secondary-structure variants run as hhv_ss_kernel");
  __shared__ float4 smem[StreamSmem<R, BT, W>::F4];
  stream_body<R, LOCAL, BT, CELLOFF, MULTI, SS, W, (FIRSTP ? 3 : 0)>(a, (int)blockIdx.x, smem);
}

// The ...AndSS variants of the 64-lane arrays (par.ssm = 2 is the reference's default, src/hhdecl.cpp:82): ONE
// workgroup of SS_WAVES = 8 independent wavefronts per CU - each a systolic array of its own with its own ring, drawing its own
// segments from the queue - sharing one LDS copy of the premultiplied score table (ssw * S33: 44 x 44 floats = 7.6 KB; eight
// one-wave workgroups with a copy each would not fit: 8 x (14 + 7.6) KB > 160 KB).  Eight bodies, one table: 8 x 14 KB (19 KB with
// LDS-parked query rows) + 7.6 KB <= 160 KB, 2 waves per SIMD as everywhere.
constexpr int SS_WAVES = 8;
constexpr int SS_TAB_FLOATS = 4 * 11 * 4 * 11;  // the largest of the three tables (S33)
template <int R, bool LOCAL, bool BT, bool CELLOFF, bool MULTI, bool FIRSTP = false>
__global__ void __launch_bounds__(SS_WAVES * LANES, 2) hhv_ss_kernel(StreamArgs a) {
  constexpr int F4 = StreamSmem<R, BT, LANES>::F4;
  __shared__ float4 smem[SS_WAVES * F4];
  __shared__ float tab[SS_TAB_FLOATS];
  static_assert(SS_TAB_FLOATS <= 4 * SS_WAVES * LANES, "table copy: four entries per thread");
#pragma unroll
  for (int k = 0; k < 4; ++k) {  // (no loop: tools/audit_asm.py reads everything behind the first loop header as the step loop)
    const int e = (int)threadIdx.x + k * SS_WAVES * LANES;
    if (e < a.ss_tab_n) tab[e] = a.ss_table[e];
  }
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  stream_body<R, LOCAL, BT, CELLOFF, MULTI, true, LANES, (FIRSTP ? 3 : 0)>(a, (int)blockIdx.x * SS_WAVES + wv, smem + wv * F4, lds_addr_of(tab));
}

// Two strips of R0 and R1 rows per lane as the two wavefronts of one workgroup (PairLds above).  Which wave index takes which
// strip alternates with the workgroup number: the strips differ in work (R0 >= R1), and the waves of the workgroups that share a
// SIMD should not all be the heavy ones.
// CHAIN: queries of more than two strips are a sequence of such launches (and of single-strip launches where no pair kernel
// exists); bit 0 = this pair is not the first link (its first strip takes its carry row from HBM: a.carry, a.results),
// bit 1 = not the last (its second strip leaves its bottom row and running best there).  a.row_base / a.qpack / a.bt_plane are
// those of the pair's first strip.
template <int R0, int R1, bool LOCAL, bool BT, int CHAIN = 0>
__global__ void __launch_bounds__(2 * LANES, 2) hhv_pair_kernel(StreamArgs a) {
  PairLds* const p = pair_lds();
  if (threadIdx.x == 0) {
    p->seg_count = 0;
    p->w0_done = 0;
    p->w1_done = 0;
  }
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int swap = a.pair_swap >= 0 ? ((int)blockIdx.x >> a.pair_swap) & 1 : 0;
  if ((wv ^ swap) == 0) {
    StreamArgs a0 = a;
    a0.pass_first = (CHAIN & 1) ? 0 : 1;
    a0.pass_last = 0;
    __shared__ float4 smem0[StreamSmem<R0, BT, LANES>::F4];
    stream_body<R0, LOCAL, BT, false, true, false, LANES, ((CHAIN & 1) ? 4 : 1)>(a0, (int)blockIdx.x, smem0, 0, p);
  } else {
    StreamArgs a1 = a;
    a1.row_base = a.row_base + LANES * R0;
    a1.qpack = a.qpack + (size_t)(LANES * R0) * REC_DW;
    a1.bt_plane = a.bt_plane + 1;
    a1.pass_first = 0;
    a1.pass_last = (CHAIN & 2) ? 0 : 1;
    __shared__ float4 smem1[StreamSmem<R1, BT, LANES>::F4];
    stream_body<R1, LOCAL, BT, false, true, false, LANES, ((CHAIN & 2) ? 5 : 2)>(a1, (int)blockIdx.x, smem1, 0, p);
  }
}

// The same for the ...AndSS variants: FOUR pairs per workgroup (eight wavefronts, like hhv_ss_kernel) around one LDS copy of the
// score table: 4 x (8 KB FIFO + 2 x 14 KB rings) + 7.6 KB = 152 KB.  Pair number = wavefront / 2; the pairs are independent of
// each other (each has its PairLds, draws its own segments), the two waves of a pair work together exactly as in hhv_pair_kernel.
constexpr int SS_PAIRS = SS_WAVES / 2;
template <int R0, int R1, bool LOCAL, bool BT, int CHAIN = 0>
__global__ void __launch_bounds__(SS_WAVES * LANES, 2) hhv_ss_pair_kernel(StreamArgs a) {
  constexpr int F0 = StreamSmem<R0, BT, LANES>::F4, F1 = StreamSmem<R1, BT, LANES>::F4;
  __shared__ PairLds pl[SS_PAIRS];
  __shared__ float4 smem0[SS_PAIRS * F0];
  __shared__ float4 smem1[SS_PAIRS * F1];
  __shared__ float tab[SS_TAB_FLOATS];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = (int)threadIdx.x + k * SS_WAVES * LANES;
    if (e < a.ss_tab_n) tab[e] = a.ss_table[e];
  }
  if (threadIdx.x < SS_PAIRS) {
    pl[threadIdx.x].seg_count = 0;
    pl[threadIdx.x].w0_done = 0;
    pl[threadIdx.x].w1_done = 0;
  }
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int pw = wv >> 1, pair_no = (int)blockIdx.x * SS_PAIRS + pw;
  const int swap = a.pair_swap >= 0 ? (pair_no >> a.pair_swap) & 1 : 0;
  const uint32_t tab_addr = lds_addr_of(tab);
  if (((wv & 1) ^ swap) == 0) {
    StreamArgs a0 = a;
    a0.pass_first = (CHAIN & 1) ? 0 : 1;
    a0.pass_last = 0;
    stream_body<R0, LOCAL, BT, false, true, true, LANES, ((CHAIN & 1) ? 4 : 1)>(a0, pair_no, smem0 + pw * F0, tab_addr, &pl[pw]);
  } else {
    StreamArgs a1 = a;
    a1.row_base = a.row_base + LANES * R0;
    a1.qpack = a.qpack + (size_t)(LANES * R0) * REC_DW;
    a1.bt_plane = a.bt_plane + 1;
    a1.pass_first = 0;
    a1.pass_last = (CHAIN & 2) ? 0 : 1;
    stream_body<R1, LOCAL, BT, false, true, true, LANES, ((CHAIN & 2) ? 5 : 2)>(a1, pair_no, smem1 + pw * F1, tab_addr, &pl[pw]);
  }
}
// ---- kernel selection (instantiates the variants of one W in the including unit) --------------------------------------
// (ss with 64-lane arrays: workgroups of SS_WAVES wavefronts, hhv_ss_kernel - the launcher asks stream_kernel_waves())
template <int W, int R, bool LOCAL, bool BT, bool CELLOFF, bool SS>
struct StreamKernelOf {
  template <bool MULTI, bool FIRSTP>
  static void* get() { return (void*)hhv_stream_kernel<R, LOCAL, BT, CELLOFF, MULTI, SS, W, FIRSTP>; }
};
template <int R, bool LOCAL, bool BT, bool CELLOFF>
struct StreamKernelOf<LANES, R, LOCAL, BT, CELLOFF, true> {
  template <bool MULTI, bool FIRSTP>
  static void* get() { return (void*)hhv_ss_kernel<R, LOCAL, BT, CELLOFF, MULTI, FIRSTP>; }
};
template <int W, int R, bool LOCAL, bool BT, bool CELLOFF>
static void* stream_kernel_ptr(bool multi, bool ss, bool first_strip) {
  if (W == LANES && multi && first_strip)
    return ss ? StreamKernelOf<W, R, LOCAL, BT, CELLOFF, true>::template get<(W == LANES), (W == LANES)>()
              : StreamKernelOf<W, R, LOCAL, BT, CELLOFF, false>::template get<(W == LANES), (W == LANES)>();
  if (W == LANES && multi)
    return ss ? StreamKernelOf<W, R, LOCAL, BT, CELLOFF, true>::template get<(W == LANES), false>()
              : StreamKernelOf<W, R, LOCAL, BT, CELLOFF, false>::template get<(W == LANES), false>();
  return ss ? StreamKernelOf<W, R, LOCAL, BT, CELLOFF, true>::template get<false, false>()
            : StreamKernelOf<W, R, LOCAL, BT, CELLOFF, false>::template get<false, false>();
}
template <int W, int R>
static void* stream_kernel_variant(bool local, bool bt, bool celloff, bool multi, bool ss, bool first_strip) {
  if (celloff)
    return local ? stream_kernel_ptr<W, R, true, true, true>(multi, ss, first_strip) : stream_kernel_ptr<W, R, false, true, true>(multi, ss, first_strip);
  if (bt)
    return local ? stream_kernel_ptr<W, R, true, true, false>(multi, ss, first_strip) : stream_kernel_ptr<W, R, false, true, false>(multi, ss, first_strip);
  return local ? stream_kernel_ptr<W, R, true, false, false>(multi, ss, first_strip) : stream_kernel_ptr<W, R, false, false, false>(multi, ss, first_strip);
}
template <int W>
static void* stream_kernel_pick(int R, bool local, bool bt, bool celloff, bool multi, bool ss, bool first_strip = false) {
  switch (R) {
    case 1: return stream_kernel_variant<W, 1>(local, bt, celloff, multi, ss, first_strip);
    case 2: return stream_kernel_variant<W, 2>(local, bt, celloff, multi, ss, first_strip);
    case 3: return stream_kernel_variant<W, 3>(local, bt, celloff, multi, ss, first_strip);
    case 4: return stream_kernel_variant<W, 4>(local, bt, celloff, multi, ss, first_strip);
    case 5: return stream_kernel_variant<W, 5>(local, bt, celloff, multi, ss, first_strip);
  }
  return nullptr;
}

}  // namespace hhv
