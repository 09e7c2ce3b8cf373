// viterbi_runner.h -- C++ host layer ABOVE the C ABI: the MI355X counterpart of the reference's
// ViterbiRunner (src/hhviterbirunner.h:50-58, src/hhviterbirunner.cpp:75-210).  It talks to the GPU
// only through include/hhviterbi_hip.h; it contains no DP arithmetic.
//
// Same control flow as ViterbiRunner::alignment:
//   for alignment = 0 .. par.altali-1                                   (:104)
//     align every template still in the work list                       (:122-168)
//       - templates that already produced hits are masked with the +-40 cross around every earlier
//         path (exclude_alignments -> Viterbi::ExcludeAlignment, :152-155,273-289; hhviterbi.cpp:61-77)
//     one Hit per template with irep = alignment+1                      (:35-62, :257)
//     templates whose Hit.score > par.smin enter the next round         (:260-268)
// Differences, by design: templates are uploaded once and stay resident (no per-batch MapHMMVector),
// a round is one kernel launch instead of n/8 Align calls, hits are returned in template order per
// round (the reference's order depends on OpenMP scheduling), errors are exceptions of type
// hhv::Error carrying the C ABI status instead of exit().
#pragma once
#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hhviterbi_hip.h"

namespace hhv {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

// the members of the reference's Parameters that the Viterbi stage consumes
// (src/hhviterbirunner.h:32-33 and src/hhviterbirunner.cpp:104,260; defaults src/hhdecl.cpp:86-98)
struct Parameters {
  int loc = 1;          // par.loc
  float egq = 0.0f;     // par.egq
  float egt = 0.0f;     // par.egt
  float shift = -0.03f; // par.shift
  float corr = 0.1f;    // par.corr
  float ssw = 0.11f;    // par.ssw
  int ssm = 2;          // par.ssm
  int altali = 4;       // par.altali
  float smin = 20.0f;   // par.smin
  std::string exclstr;           // par.exclstr          ("-excl 10-20,40-55": query rows never aligned)
  std::string template_exclstr;  // par.template_exclstr (same for template columns)
};

// a prepared profile HMM (after PrepareQueryHMM / PrepareTemplateHMM): p[(L+1)*20], tr[(L+1)*7]
struct Profile {
  int L = 0;
  const float* p = nullptr;
  const float* tr = nullptr;
};

// the Hit fields ViterbiConsumerThread::align fills (src/hhviterbirunner.cpp:35-62)
struct Hit {
  int entry = -1;  // index of the template in the input list (reference: HHEntry* entry)
  int irep = 0;    // index of the alternative alignment, 1-based (:257)
  int lastrep = 0; // score <= smin (:37)
  float score = 0, score_ss = 0, score_aass = 0;
  int i1 = 0, j1 = 0, i2 = 0, j2 = 0, nsteps = 0, matched_cols = 0;
  std::vector<int32_t> i, j;   // 1-based path arrays, index 0 unused, step 1 = alignment end
  std::vector<int8_t> states;
  std::vector<float> S, S_ss;
};

// The ss_hmm_mode ViterbiConsumerThread::align derives for a batch (src/hhviterbirunner.cpp:14-22), restated
// literally: consensus = AND over the batch of HMM::computeScoreSSMode(q, t) (src/hhhmm.cpp:1967-1973); the
// selection chain can only ever return 0 or PRED_PRED (4) - PRED_DSSP / DSSP_PRED are discarded.
inline int SelectSSMode(int consensus_ss_hmm_mode) {
  int ss_hmm_mode = (consensus_ss_hmm_mode & 1 /*PRED_DSSP*/);
  ss_hmm_mode = (ss_hmm_mode == 0) ? consensus_ss_hmm_mode & 2 /*DSSP_PRED*/ : 0;
  ss_hmm_mode = (ss_hmm_mode == 0) ? consensus_ss_hmm_mode & 4 /*PRED_PRED*/ : 0;
  return ss_hmm_mode;
}

// Viterbi::ExcludeAlignment (src/hhviterbi.cpp:61-77): OR the +-VITERBI_PATH_WIDTH cross of one path
// into mask[(Lq+1)*(Lt+1)]
void ExcludeAlignment(std::vector<uint8_t>& mask, int Lq, int Lt, const int32_t* i_steps, const int32_t* j_steps,
                      int nsteps);

// ViterbiRunner::exclude_regions / exclude_template_regions (src/hhviterbirunner.cpp:291-329): every pair of
// integers found in the string (parsed like strint, src/util.cpp:133-151, sign ignored) switches off the query
// rows i0..i1 (resp. template columns j0..j1) of mask[(Lq+1)*(Lt+1)]
void ExcludeRegions(std::vector<uint8_t>& mask, int Lq, int Lt, const std::string& exclstr);
// the (lo, hi) pairs the two functions iterate over (lo = max(1, |a|), hi = |b|; the caller clips hi to the length)
std::vector<int32_t> ParseRegions(const std::string& exclstr);
void ExcludeTemplateRegions(std::vector<uint8_t>& mask, int Lq, int Lt, const std::string& exclstr);

class ViterbiRunner {
 public:
  explicit ViterbiRunner(int device = 0) : device_(device) {}
  // fetch_paths: also copy the per-hit path arrays (needed for alt rounds > 1 anyway)
  std::vector<Hit> alignment(const Parameters& par, const Profile& q, const std::vector<Profile>& templates);

 private:
  int device_;
};

// The same search with the template database sharded over several GPUs of one node.  The reference's unit
// of independence is the batch loop of ViterbiRunner::alignment (src/hhviterbirunner.cpp:122: `#pragma omp parallel for`
// over SIMD batches, every batch aligned on its own; the results are merged serially afterwards, :173): that is where
// the shard boundary goes.  Whole templates are distributed (hhv_shard_plan: length-sorted bins of 64, longest
// processing time first), every shard runs ALL alternative-alignment rounds of its templates on its own device with its
// own hhv_ctx (rounds >= 2 only touch the templates and paths of the same shard: no exchange), one host thread per
// shard, and the hit lists are merged in the order a single ViterbiRunner returns them (round, then template order).
//   devices: one entry per shard; a device may be listed more than once (logical shards on one GPU - what the tests use
//            on a one-GPU box).  Empty = one shard per device of the node (hhv_device_count).
class ShardedViterbiRunner {
 public:
  explicit ShardedViterbiRunner(const std::vector<int>& devices = std::vector<int>());
  std::vector<Hit> alignment(const Parameters& par, const Profile& q, const std::vector<Profile>& templates);
  const std::vector<int>& devices() const { return devices_; }
  // templates per shard of the last call (for reporting / tests)
  const std::vector<int>& shard_sizes() const { return shard_sizes_; }

 private:
  std::vector<int> devices_;
  std::vector<int> shard_sizes_;
};

}  // namespace hhv
