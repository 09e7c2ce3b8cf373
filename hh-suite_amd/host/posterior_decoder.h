// posterior_decoder.h -- host side of the MAC realignment above the C ABI (hhv_mac_realign):
// the MI355X counterpart of
//   PosteriorDecoderRunner::executeComputation   src/hhposteriordecoderrunner.cpp:43-119  (grouping by template, rounds)
//   PosteriorDecoder::realign                    src/hhposteriordecoder.cpp:86-119        (mask construction around the DP)
//     initializeForAlignment / maskViterbiAlignment / excludeMACAlignment   :151-262
//     exclude_regions / exclude_template_regions                            :121-149
// It contains no DP arithmetic: it builds the cell-off masks and batches the hits; forward / backward / MAC /
// backtrace run on the GPU, one wavefront per hit, all hits of a round in one launch.
//
// Control flow of the reference: hits are grouped by template; the alternative alignments of ONE template are
// realigned in the order of their irep, each excluding the cells (+-2) of the MAC alignments found before it; different
// templates are independent (OpenMP loop).  Here round r realigns the r-th hit of every template in one batch.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "viterbi_runner.h"

namespace hhv {

// 2^tr for the seven transitions (HMM::Log2LinTransitionProbs, src/hhhmm.cpp:2305-2313: powf) plus the boundary
// assignments the realign stage makes: query = initializeQueryHMMTransitions (src/hhposteriordecoderrunner.cpp:146-155),
// template = initializeForAlignment (src/hhposteriordecoder.cpp:159-167).  out[(L+1)*7]
void LinearTransitions(const float* tr_log2, int L, bool is_query, float* out);

struct MacParameters {
  int loc = 1;            // par.loc
  float shift = -0.03f;   // par.shift
  float mact = 0.3501f;   // par.mact (src/hhdecl.cpp:97)
  int min_overlap = 0;    // par.min_overlap
  std::string exclstr, template_exclstr;
};

// what realign() reads of a Viterbi hit (src/hhposteriordecoder.cpp:205-240)
struct MacInput {
  int entry = -1;  // template index (into `templates`)
  int resident = -1;  // with useResidentSet(): index of this template in the resident set (-1: same as entry)
  int irep = 1;    // rank of this alignment among the template's alternatives
  int i1 = 0, j1 = 0, i2 = 0, j2 = 0, nsteps = 0;
  const int32_t* i = nullptr;  // Viterbi path, entries 1..nsteps; with useResidentSet(): i == j == nullptr = take end points
  const int32_t* j = nullptr;  // and path of template `entry` from the set's own hhv_hits results
};

// the Hit fields backtraceMAC fills (src/hhbacktracemac.cpp:113-240); score / P-values are restored by realign()
struct MacAlignment {
  int entry = -1, irep = 1;
  double Pforward = 0;
  float sum_of_probs = 0;
  int i1 = 0, j1 = 0, i2 = 0, j2 = 0, nsteps = 0, matched_cols = 0;
  std::vector<int32_t> i, j;  // entries 1..nsteps; also Hit::alt_i / alt_j
  std::vector<int8_t> states;
  std::vector<float> S, P_posterior;
};

// the mask realign() builds before the DP: InitializeForAlignment (non-self) + maskViterbiAlignment + every
// earlier MAC alignment of the template (+-2 cross) + -excl regions.  mask[(Lq+1)*(Lt+1)], 1 = cell off
void MacCellOff(int Lq, int Lt, const MacParameters& par, const MacInput& hit, const std::vector<const MacAlignment*>& earlier,
                std::vector<uint8_t>* mask);

class PosteriorDecoderRunner {
 public:
  explicit PosteriorDecoderRunner(hhv_ctx* ctx) : ctx_(ctx), resident_(nullptr) {}
  // The templates are those of a resident template set (the one the Viterbi stage searched; MacInput::entry = index in
  // it): their profiles are read on the device, Profile::p of `templates` is not used, only L and the linear tr.
  void useResidentSet(hhv_tset* ts) { resident_ = ts; }
  // q / templates: prepared profiles with LINEAR transitions (LinearTransitions above); hits in any order.
  // Returns one MacAlignment per input hit, in input order.  Throws hhv::Error.
  std::vector<MacAlignment> executeComputation(const MacParameters& par, const Profile& q, const std::vector<Profile>& templates,
                                               const std::vector<MacInput>& hits);

 private:
  hhv_ctx* ctx_;
  hhv_tset* resident_;
};

}  // namespace hhv

extern "C" {
// plain-C shim for bindings/tests.  Hits: n_hits rows of 8 ints (entry, irep, i1, j1, i2, j2, nsteps, resident) + concatenated paths
// (path_off[n_hits+1], entries 1..nsteps at path_off[k]+1 ...).  Outputs: out_scalars[n_hits][6] = nsteps,i1,j1,i2,j2,
// matched_cols; out_real[n_hits][2] = Pforward, sum_of_probs; paths into out_i/out_j/out_states/out_S/out_P with row pitch pcap.
/* resident: nullable; when given, t_p is ignored and the profiles of template set `resident` are used (entry = index in it) */
int hhvr_mac_realign(hhv_ctx* ctx, hhv_tset* resident, int32_t loc, float shift, float mact, int32_t min_overlap, const char* exclstr,
                     const char* template_exclstr, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n_templates,
                     const int32_t* Lt, const float* const* t_p, const float* const* t_tr_lin, int32_t n_hits,
                     const int32_t* hit_rows, const int64_t* path_off, const int32_t* path_i, const int32_t* path_j,
                     int32_t* out_scalars, double* out_real, int32_t pcap, int32_t* out_i, int32_t* out_j, int8_t* out_states,
                     float* out_S, float* out_P);
void hhvr_mac_last_timing(double* ms3 /* host masks, hhv_mac_realign, path fetch of the last run */);
void hhvr_linear_transitions(const float* tr_log2, int32_t L, int32_t is_query, float* out);
int hhvr_mac_celloff(int32_t Lq, int32_t Lt, int32_t min_overlap, const char* exclstr, const char* template_exclstr,
                     const int32_t* hit_row, const int32_t* path_i, const int32_t* path_j, int32_t n_prev,
                     const int32_t* prev_off, const int32_t* prev_i, const int32_t* prev_j, uint8_t* mask);
}
