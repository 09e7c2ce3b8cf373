// prefilter.h -- host side of the HHblits prefilter, mirroring the reference's class
//   Prefilter                      src/hhprefilter.h:72-127
//   Prefilter::prefilter_db        src/hhprefilter.cpp:428-596   two-stage selection around the two kernels
//   Prefilter::stripe_query_profile src/hhprefilter.cpp:355-425  query HMM x 219 context states -> byte profile
// above the C ABI (hhv_prefilter_upload_db / hhv_prefilter_scores): the database of column-state sequences is
// uploaded once and stays in HBM; per query the two kernels run on the GPU and only the thresholding, the two
// sorts and the e-value arithmetic - O(n_db) scalar work - stay on the host.
#ifndef HHV_HOST_PREFILTER_H_
#define HHV_HOST_PREFILTER_H_

#include <stdint.h>

#include <string>
#include <vector>

#include "hhviterbi_hip.h"

namespace hhv {

// util-inl.h:83-93 / :190-214 -- the reference's polynomial log2 / 2^x (they define scores and e-values)
float flog2(float x);
float fpow2(float x);

// hhblits defaults: src/hhdecl.cpp:118-127 (prefilter_*), -maxfilt 20000
struct PrefilterParams {
  int gap_open = 20, gap_extend = 4, score_offset = 50, bit_factor = 4;
  double evalue_thresh = 1000.0, evalue_coarse_thresh = 100000.0;
  int smax_thresh = 10, min_hits = 100, maxnumdb = 20000;
};

// The 219 context states of a cs219 library file (data/cs219.lib; cs::ContextLibrary<AA>(FILE*) +
// TransformToLin, src/cs/context_profile-inl.h:81-144): central-column probabilities [219][20] as doubles.
// Returns false (and *err) on malformed input.
bool ReadContextLibrary(const std::string& path, std::vector<double>* probs, std::string* err);

// stripe_query_profile without the striping (the kernel stripes in LDS): q_p = the rows p[0..Lq-1][20] of the
// prefilter query HMM (the reference indexes from 0, :364-370), pav[20], lib[219][20]; plain[220][Lq] bytes.
void PrefilterQueryProfile(const float* q_p, const float* pav, const double* lib, int Lq, int score_offset, int bit_factor,
                           uint8_t* plain);

class Prefilter {
 public:
  // uploads the database (n_db sequences concatenated, offsets[n_db+1]); lib[219][20] is copied
  Prefilter(hhv_ctx* ctx, int32_t n_db, const uint8_t* seqs, const int64_t* offsets, const double* lib);
  ~Prefilter();
  bool ok() const { return db_ != nullptr; }

  // ids of the sequences that pass both filters, in the order the reference emits them; evalues (optional) aligned
  // to it.  Returns a negative hhv_status on failure.
  int prefilter_db(const float* q_p, const float* q_pav, int Lq, const PrefilterParams& par, std::vector<int32_t>* selected,
                   std::vector<double>* evalues = nullptr, int* passed_first = nullptr);

  // the two host-side selection steps of prefilter_db on given kernel scores (pure functions; also what the CPU
  // tests drive): ungapped[n_db] -> subset (ids, best first);  sw[n_subset] (aligned to subset) -> selected ids
  static void SelectFirst(const int32_t* ungapped, const int32_t* length, int n_db, int Lq, const PrefilterParams& par,
                          std::vector<int32_t>* subset);
  static void SelectSecond(const int32_t* sw, const int32_t* subset, int n_subset, const int32_t* length, int n_db, int Lq,
                           const PrefilterParams& par, std::vector<int32_t>* selected, std::vector<double>* evalues);

 private:
  hhv_ctx* ctx_;
  hhv_pfdb* db_;
  std::vector<int32_t> length_;
  std::vector<double> lib_;
};

}  // namespace hhv

extern "C" {
// plain-C shim for bindings/tests: one-shot prefilter_db on a database handed over as host arrays
int hhvr_prefilter_db(hhv_ctx* ctx, int32_t n_db, const uint8_t* seqs, const int64_t* offsets, const double* lib,
                      const float* q_p, const float* q_pav, int32_t Lq, const int32_t* ipar /* gap_open, gap_extend,
                      score_offset, bit_factor, smax_thresh, min_hits, maxnumdb */, const double* dpar /* evalue_thresh,
                      evalue_coarse_thresh */, int32_t* out_ids, double* out_evalues, int32_t out_cap, int32_t* passed_first);
int hhvr_prefilter_select_first(const int32_t* ungapped, const int32_t* length, int32_t n_db, int32_t Lq, const int32_t* ipar,
                                const double* dpar, int32_t* subset);
int hhvr_prefilter_select_second(const int32_t* sw, const int32_t* subset, int32_t n_subset, const int32_t* length, int32_t n_db,
                                 int32_t Lq, const int32_t* ipar, const double* dpar, int32_t* out_ids, double* out_evalues);
int hhvr_prefilter_profile(const float* q_p, const float* pav, const double* lib, int32_t Lq, int32_t score_offset,
                           int32_t bit_factor, uint8_t* plain);
int hhvr_read_context_library(const char* path, double* probs /* [219][20] */);
float hhvr_flog2(float x);
float hhvr_fpow2(float x);
}

#endif  // HHV_HOST_PREFILTER_H_
