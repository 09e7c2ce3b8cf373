// tests/emul/wave_emul.cpp -- TEST INFRASTRUCTURE (never part of the product library).
//
// Host-side lock-step emulation of ONE wavefront of the systolic Viterbi kernel: the same
// per-lane code (hh-suite_amd/csrc/viterbi_lane.h) is stepped for 64 lanes, with the DPP
// wave_shr:1 exchange replaced by reading lane g-1's state before it is updated (lanes are
// visited 63 -> 0 inside a step).  It lets the schedule, the boundary handling and the result
// plumbing be debugged against the oracle on the CPU-only build container; GPU parity itself is
// established by the -m gpu tests through the C ABI.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../hh-suite_amd/csrc/viterbi_lane.h"

using namespace hhv;

// One pass over the wave's stream (mirrors hhv_stream_kernel, incl. the multi-pass carry of long queries).
struct Carry {
  float MM, GD, IM, DG, MI;
};

// secondary-structure inputs of the emulation (null table = no SS): the premultiplied table ssw*S, the
// table row offset of every query row and where the template index sits in the record meta
struct SSArgs {
  const float* table;
  const int32_t* q_off;  // [passes*64*R]
  int t_shift, t_mask;
};

template <int R, bool LOCAL, bool BT, bool CELLOFF>
static int run_pass(const float* qpack, const float* records, long M, Params P, TemplateResult* results, int n_results,
                    uint64_t* bt /* M x 64 entries of this pass, in/out */, int row_base, bool first, bool last,
                    std::vector<Carry>& carry, const SSArgs& ss) {
  std::vector<LaneState<R>> st(64);
  std::vector<QRows<R>> q(64);
  for (int g = 0; g < 64; ++g) {
    st[g].reset();
    q[g].load(qpack + (size_t)g * R * REC_DW);
  }
  const int g_last = last ? (P.Lq - row_base - 1) / R : 63;
  const int r_last = (P.Lq - row_base - 1) % R;
  int emitted = 0;
  for (long s = 0; s < M + 63; ++s) {
    for (int g = 63; g >= 0; --g) {
      const long r = s - g;
      if (r < 0 || r >= M) continue;
      const float* rec = records + (size_t)r * REC_DW;
      int32_t meta;
      memcpy(&meta, rec + REC_META, 4);
      const DiagSums ds = lane_diag(st[g], q[g]);  // reads the previous step's hand-off, like the kernel: before the new one
      Incoming in;
      if (g == 0) {
        in = boundary_incoming(meta, P);
        if (!first) {
          in.MM = carry[r].MM;
          in.GD = carry[r].GD;
          in.IM = carry[r].IM;
          in.DG = carry[r].DG;
          in.MI = carry[r].MI;
          if (meta < 0 && st[0].tid >= 0 && st[0].tid < n_results) {
            in.fs = results[st[0].tid].score;
            in.fpos = (results[st[0].tid].i2 << 16) | results[st[0].tid].j2;
          }
        }
      } else {
        const LaneState<R>& a = st[g - 1];
        in.MM = a.MM[R - 1];
        in.GD = a.GD[R - 1];
        in.IM = a.IM[R - 1];
        in.DG = a.DG[R - 1];
        in.MI = a.MI[R - 1];
        in.fs = a.fs;
        in.fpos = a.fpos;
      }
      const int i0 = row_base + g * R + 1;
      if (meta < 0) {
        int32_t new_tid;
        memcpy(&new_tid, rec + 0, 4);
        TemplateResult res;
        if (lane_header<R, LOCAL, true>(st[g], q[g], in, i0, new_tid, P, g == g_last, res)) {
          if (res.tid >= 0 && res.tid < n_results) results[res.tid] = res;
          emitted++;
        }
      } else {
        const int j = meta & META_JMASK;
        uint64_t cell = 0;
        if (CELLOFF) cell = bt[(size_t)r * 64 + g];
        uint64_t bytes;
        ArraySrc src;
        src.rec = rec;
        if (ss.table) {
          float ssv[R];
          const int tidx = (meta >> ss.t_shift) & ss.t_mask;
          for (int r = 0; r < R; ++r) ssv[r] = ss.table[ss.q_off[i0 - 1 + r] + tidx];
          PtrSs ssx{ssv};
          bytes = lane_column<R, LOCAL, BT, CELLOFF, true, true, bt_mm_mode(64, R, LOCAL, CELLOFF, true), bt_pair_mode(64, CELLOFF)>(st[g], q[g], in, ds, src, j, i0, r_last, P, cell, ssx);
        } else {
          PtrSs noss{nullptr};
          bytes = lane_column<R, LOCAL, BT, CELLOFF, true, false, bt_mm_mode(64, R, LOCAL, CELLOFF, false), bt_pair_mode(64, CELLOFF)>(st[g], q[g], in, ds, src, j, i0, r_last, P, cell, noss);
        }
        if (BT) bt[(size_t)r * 64 + g] = bytes;
      }
      if (!last && g == 63) {
        Carry c = {st[g].MM[R - 1], st[g].GD[R - 1], st[g].IM[R - 1], st[g].DG[R - 1], st[g].MI[R - 1]};
        carry[r] = c;
      }
    }
  }
  return emitted;
}

// qpack holds passes * 64 * R rows; bt holds passes planes of M x 64 entries
template <int R, bool LOCAL, bool BT, bool CELLOFF>
static int run_wave(const float* qpack, const float* records, long M, Params P, TemplateResult* results, int n_results,
                    uint64_t* bt, int passes, const SSArgs& ss) {
  std::vector<Carry> carry(M);
  int emitted = 0;
  for (int p = 0; p < passes; ++p) {
    emitted = run_pass<R, LOCAL, BT, CELLOFF>(qpack + (size_t)p * 64 * R * REC_DW, records, M, P, results, n_results,
                                              bt ? bt + (size_t)p * M * 64 : nullptr, p * 64 * R, p == 0, p == passes - 1,
                                              carry, ss);
  }
  return emitted;
}

template <int R>
static int dispatch(int local, int want_bt, int celloff, const float* qpack, const float* records, long M, Params P,
                    TemplateResult* results, int n_results, uint64_t* bt, int passes, const SSArgs& ss) {
  if (celloff) {
    return local ? run_wave<R, true, true, true>(qpack, records, M, P, results, n_results, bt, passes, ss)
                 : run_wave<R, false, true, true>(qpack, records, M, P, results, n_results, bt, passes, ss);
  }
  if (want_bt) {
    return local ? run_wave<R, true, true, false>(qpack, records, M, P, results, n_results, bt, passes, ss)
                 : run_wave<R, false, true, false>(qpack, records, M, P, results, n_results, bt, passes, ss);
  }
  return local ? run_wave<R, true, false, false>(qpack, records, M, P, results, n_results, bt, passes, ss)
               : run_wave<R, false, false, false>(qpack, records, M, P, results, n_results, bt, passes, ss);
}

// encoding of the MM predecessor the emulated variant writes (viterbi_lane.h bt_mm_mode): the tests decode with it
extern "C" int hhv_emul_bt_mm_mode(int R, int local, int celloff, int ss) { return bt_mm_mode(64, R, local != 0, celloff != 0, ss != 0); }

extern "C" int hhv_emul_wave(int R, int local, int want_bt, int celloff, const float* qpack, const float* records,
                             long M, float egq, float egt, float shift, int Lq, TemplateResult* results,
                             int n_results, uint64_t* bt, int passes, const float* ss_table, const int32_t* ss_q_off,
                             int ss_t_shift, int ss_t_mask) {
  SSArgs ss = {ss_table, ss_q_off, ss_t_shift, ss_t_mask};
  Params P;
  P.set_gaps(egq, egt);
  P.shift = shift;
  P.Lq = Lq;
  switch (R) {
    case 1: return dispatch<1>(local, want_bt, celloff, qpack, records, M, P, results, n_results, bt, passes, ss);
    case 2: return dispatch<2>(local, want_bt, celloff, qpack, records, M, P, results, n_results, bt, passes, ss);
    case 3: return dispatch<3>(local, want_bt, celloff, qpack, records, M, P, results, n_results, bt, passes, ss);
    case 4: return dispatch<4>(local, want_bt, celloff, qpack, records, M, P, results, n_results, bt, passes, ss);
    case 5: return dispatch<5>(local, want_bt, celloff, qpack, records, M, P, results, n_results, bt, passes, ss);
  }
  return -1;
}
