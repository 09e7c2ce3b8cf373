// prefilter.cpp -- see prefilter.h.  Host-side selection logic of the HHblits prefilter around the GPU kernels.
#include "prefilter.h"

#include <chrono>

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <utility>

namespace hhv {

// util-inl.h:83-93.  The polynomial constants are double literals there, so the Horner chain runs in double and
// is rounded to float once.
float flog2(float x) {
  if (x <= 0) return -128;
  int32_t bits;
  memcpy(&bits, &x, 4);
  const float e = (float)(((bits & 0x7F800000) >> 23) - 0x7f);
  bits = (bits & 0x007FFFFF) | 0x3f800000;
  memcpy(&x, &bits, 4);
  x = (float)((double)x - 1.0);
  x = (float)((double)x * (1.441740 + (double)x * (-0.7077702 + (double)x * (0.4123442 + (double)x * (-0.1903190 + (double)x * 0.0440047)))));
  return x + e;
}

// util-inl.h:190-214 (float constants, float arithmetic)
float fpow2(float x) {
  if (x >= FLT_MAX_EXP) return FLT_MAX;
  if (x <= FLT_MIN_EXP) return 0.0f;
  const float tx = (x - 0.5f) + (float)(3 << 22);
  int32_t lx;
  memcpy(&lx, &tx, 4);
  lx -= 0x4b400000;
  const float dx = x - (float)lx;
  float r = 1.0f + dx * (0.693019f + dx * (0.241404f + dx * (0.0520749f + dx * 0.0134929f)));
  int32_t rb;
  memcpy(&rb, &r, 4);
  rb += lx << 23;
  memcpy(&r, &rb, 4);
  return r;
}

bool ReadContextLibrary(const std::string& path, std::vector<double>* probs, std::string* err) {
  FILE* f = fopen(path.c_str(), "r");
  if (!f) {
    if (err) *err = "cannot open " + path;
    return false;
  }
  probs->clear();
  char line[4096];
  int size = -1, lib_len = -1;
  bool in_profile = false, want_row = false;
  while (fgets(line, sizeof(line), f)) {
    if (!in_profile) {
      if (!strncmp(line, "SIZE", 4)) size = atoi(line + 4);
      if (!strncmp(line, "LENG", 4) && lib_len < 0) lib_len = atoi(line + 4);
      if (!strncmp(line, "ContextProfile", 14)) in_profile = true;
      continue;
    }
    if (!strncmp(line, "ISLOG", 5) || !strncmp(line, "NAME", 4) || !strncmp(line, "PRIOR", 5) || !strncmp(line, "COLOR", 5) ||
        !strncmp(line, "LENG", 4) || !strncmp(line, "ALPH", 4))
      continue;  // linear or log: after TransformToLin both hold probabilities
    if (!strncmp(line, "PROBS", 5)) {
      want_row = true;
      continue;
    }
    if (line[0] == '/' && line[1] == '/') {
      in_profile = false;
      continue;
    }
    if (!strncmp(line, "ContextProfile", 14)) continue;
    if (want_row) {
      // "<column>\t<20 scaled negative log2 probabilities>"; only the central (= only) column of a length-1 library
      char* p = line;
      const long col = strtol(p, &p, 10);
      if (col == (lib_len + 1) / 2) {
        for (int a = 0; a < 20; ++a) {
          while (*p == ' ' || *p == '\t') ++p;
          if (*p == '*') {
            probs->push_back(0.0);
            ++p;
          } else {
            const long v = strtol(p, &p, 10);
            probs->push_back(pow(2.0, (double)(-v) / 1000.0));  // kScale = 1000, src/cs/globals.h:32
          }
        }
      }
    }
  }
  fclose(f);
  if (size != 219 || probs->size() != (size_t)219 * 20) {
    if (err) *err = "not a 219-state context library: " + path;
    return false;
  }
  return true;
}

void PrefilterQueryProfile(const float* q_p, const float* pav, const double* lib, int Lq, int score_offset, int bit_factor,
                           uint8_t* plain) {
  for (int i = 0; i < Lq; ++i) {
    for (int k = 0; k < 219; ++k) {
      // :364-369: float accumulator, double terms (the library holds doubles)
      float sum = 0;
      for (int a = 0; a < 20; ++a) sum = (float)((double)sum + ((double)q_p[i * 20 + a] * lib[k * 20 + a]) / (double)pav[a]);
      // :396-403
      const float dummy = (float)((double)(flog2(sum) * (float)bit_factor + (float)score_offset) + 0.5);
      uint8_t v;
      if (dummy > 255.0)
        v = 255;
      else if (dummy < 0)
        v = 0;
      else
        v = (uint8_t)dummy;
      plain[(size_t)k * Lq + i] = v;
    }
    plain[(size_t)219 * Lq + i] = (uint8_t)(score_offset - 1);  // the ANY state, :418
  }
}

Prefilter::Prefilter(hhv_ctx* ctx, int32_t n_db, const uint8_t* seqs, const int64_t* offsets, const double* lib)
    : ctx_(ctx), db_(nullptr) {
  if (hhv_prefilter_upload_db(ctx, n_db, seqs, offsets, &db_) != HHV_OK) {
    db_ = nullptr;
    return;
  }
  length_.resize(n_db);
  for (int n = 0; n < n_db; ++n) length_[n] = (int32_t)(offsets[n + 1] - offsets[n]);
  lib_.assign(lib, lib + 219 * 20);
}

Prefilter::~Prefilter() { hhv_prefilter_free_db(db_); }

namespace {
// comparePair (src/hhprefilter.cpp:14-27) takes std::pair<int,int>: the (evalue, id) pairs of the second stage are
// converted on the way in, i.e. they are ordered by the e-value TRUNCATED to int, then by id.
struct ByIntKey {
  bool operator()(const std::pair<int, int>& l, const std::pair<int, int>& r) const {
    if (l.first != r.first) return l.first < r.first;
    return l.second < r.second;
  }
};
}  // namespace

void Prefilter::SelectFirst(const int32_t* ungapped, const int32_t* length, int n_db, int Lq, const PrefilterParams& par,
                            std::vector<int32_t>* subset) {
  // :461-505: length correction, descending sort, keep everything above smax_thresh but at least min_hits
  const float log_qlen = flog2((float)Lq);
  std::vector<std::pair<int, int> > first(n_db);
  for (int n = 0; n < n_db; ++n)
    first[n] = std::make_pair(ungapped[n] - (int)((float)par.bit_factor * (log_qlen + flog2((float)length[n]))), n);
  // The reference sorts everything (descending: std::sort + std::reverse) and cuts at the first element that is beyond
  // min_hits AND not above smax_thresh, i.e. it keeps the min_hits best plus everything above the threshold, best first.
  // The same set and order without sorting the whole database: nth_element for the rank boundary, sort what is kept.
  struct Desc {
    bool operator()(const std::pair<int, int>& l, const std::pair<int, int>& r) const { return ByIntKey()(r, l); }
  };
  const size_t top = std::min<size_t>((size_t)std::max(par.min_hits, 0), first.size());
  if (top < first.size()) std::nth_element(first.begin(), first.begin() + top, first.end(), Desc());
  size_t keep = top;
  for (size_t k = top; k < first.size(); ++k)
    if (first[k].first > par.smax_thresh) std::swap(first[keep++], first[k]);
  first.resize(keep);
  std::sort(first.begin(), first.end(), Desc());
  subset->resize(keep);
  for (size_t k = 0; k < keep; ++k) (*subset)[k] = first[k].second;
}

void Prefilter::SelectSecond(const int32_t* sw, const int32_t* subset, int n_subset, const int32_t* length, int n_db, int Lq,
                             const PrefilterParams& par, std::vector<int32_t>* selected, std::vector<double>* evalues) {
  // :512-596: e-value, coarse threshold, sort, fine threshold (at least min_hits), at most maxnumdb
  const double factor = (double)n_db * Lq;
  std::vector<std::pair<int, int> > key;  // ((int) evalue, id) -- the order comparePair sees
  std::vector<double> ev;                 // e-value of key[k] before sorting, looked up by position in subset
  std::vector<int> pos_of(n_db, -1);
  ev.reserve(n_subset);
  for (int k = 0; k < n_subset; ++k) {
    const int n = subset[k];
    const double evalue = factor * length[n] * fpow2((float)(-sw[k] / par.bit_factor));
    if (evalue < par.evalue_coarse_thresh) {
      key.push_back(std::make_pair((int)evalue, n));
      pos_of[n] = (int)ev.size();
      ev.push_back(evalue);
    }
  }
  std::sort(key.begin(), key.end(), ByIntKey());
  size_t keep = key.size();
  for (size_t k = 0; k < key.size(); ++k)
    if ((int)k >= par.min_hits && ev[pos_of[key[k].second]] > par.evalue_thresh) {
      keep = k;
      break;
    }
  selected->clear();
  if (evalues) evalues->clear();
  int count = 0;
  for (size_t k = 0; k < keep; ++k) {
    ++count;
    selected->push_back(key[k].second);
    if (evalues) evalues->push_back(ev[pos_of[key[k].second]]);
    if (count >= par.maxnumdb) break;
  }
}

int Prefilter::prefilter_db(const float* q_p, const float* q_pav, int Lq, const PrefilterParams& par,
                            std::vector<int32_t>* selected, std::vector<double>* evalues, int* passed_first) {
  selected->clear();
  if (evalues) evalues->clear();
  if (!db_) return HHV_E_ARG;
  const bool timing = getenv("HHV_API_TIMING") != nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  const int n_db = (int)length_.size();
  std::vector<uint8_t> plain((size_t)220 * Lq);
  PrefilterQueryProfile(q_p, q_pav, lib_.data(), Lq, par.score_offset, par.bit_factor, plain.data());
  const double ms_profile = since(t0);
  const auto t1 = std::chrono::steady_clock::now();
  // stage 1 entirely on the GPU: gapless score of every sequence, length correction, sort, cut (SelectFirst is the
  // host statement of the same rule); only the surviving ids come back
  std::vector<int32_t> subset(n_db);
  int32_t n_sub = 0;
  int rc = hhv_prefilter_first(ctx_, db_, plain.data(), Lq, par.score_offset, flog2((float)Lq), par.bit_factor, par.smax_thresh,
                               par.min_hits, subset.data(), n_db, &n_sub);
  if (rc != HHV_OK) return rc;
  subset.resize(n_sub);
  const double ms_first = since(t1);
  const auto t2 = std::chrono::steady_clock::now();
  if (passed_first) *passed_first = (int)subset.size();
  if (subset.empty()) return HHV_OK;
  // stage 2: Smith-Waterman of the survivors on the GPU
  std::vector<int32_t> sw(subset.size());
  rc = hhv_prefilter_scores(ctx_, db_, plain.data(), Lq, par.score_offset, 1, par.gap_open + par.gap_extend, par.gap_extend,
                            subset.data(), (int32_t)subset.size(), sw.data());
  if (rc != HHV_OK) return rc;
  const double ms_sw = since(t2);
  const auto t3 = std::chrono::steady_clock::now();
  SelectSecond(sw.data(), subset.data(), (int)subset.size(), length_.data(), n_db, Lq, par, selected, evalues);
  if (timing)
    fprintf(stderr, "hhv::Prefilter::prefilter_db: %d sequences, Lq %d: profile %.2f ms, first stage (device) %.2f ms -> %d, "
            "Smith-Waterman %.2f ms, selection %.2f ms -> %zu\n", n_db, Lq, ms_profile, ms_first, n_sub, ms_sw, since(t3), selected->size());
  return HHV_OK;
}

}  // namespace hhv

extern "C" {

static hhv::PrefilterParams params_from(const int32_t* ipar, const double* dpar) {
  hhv::PrefilterParams par;
  par.gap_open = ipar[0];
  par.gap_extend = ipar[1];
  par.score_offset = ipar[2];
  par.bit_factor = ipar[3];
  par.smax_thresh = ipar[4];
  par.min_hits = ipar[5];
  par.maxnumdb = ipar[6];
  par.evalue_thresh = dpar[0];
  par.evalue_coarse_thresh = dpar[1];
  return par;
}

int hhvr_prefilter_db(hhv_ctx* ctx, int32_t n_db, const uint8_t* seqs, const int64_t* offsets, const double* lib,
                      const float* q_p, const float* q_pav, int32_t Lq, const int32_t* ipar, const double* dpar,
                      int32_t* out_ids, double* out_evalues, int32_t out_cap, int32_t* passed_first) {
  hhv::Prefilter pf(ctx, n_db, seqs, offsets, lib);
  if (!pf.ok()) return HHV_E_ARG;
  const hhv::PrefilterParams par = params_from(ipar, dpar);
  std::vector<int32_t> sel;
  std::vector<double> ev;
  int pass1 = 0;
  const int rc = pf.prefilter_db(q_p, q_pav, Lq, par, &sel, &ev, &pass1);
  if (rc != HHV_OK) return rc;
  if (passed_first) *passed_first = pass1;
  const int n = (int)std::min<size_t>(sel.size(), (size_t)out_cap);
  for (int k = 0; k < n; ++k) {
    out_ids[k] = sel[k];
    if (out_evalues) out_evalues[k] = ev[k];
  }
  return n;
}

int hhvr_prefilter_select_first(const int32_t* ungapped, const int32_t* length, int32_t n_db, int32_t Lq, const int32_t* ipar,
                                const double* dpar, int32_t* subset) {
  std::vector<int32_t> s;
  hhv::Prefilter::SelectFirst(ungapped, length, n_db, Lq, params_from(ipar, dpar), &s);
  memcpy(subset, s.data(), s.size() * sizeof(int32_t));
  return (int)s.size();
}

int hhvr_prefilter_select_second(const int32_t* sw, const int32_t* subset, int32_t n_subset, const int32_t* length, int32_t n_db,
                                 int32_t Lq, const int32_t* ipar, const double* dpar, int32_t* out_ids, double* out_evalues) {
  std::vector<int32_t> sel;
  std::vector<double> ev;
  hhv::Prefilter::SelectSecond(sw, subset, n_subset, length, n_db, Lq, params_from(ipar, dpar), &sel, &ev);
  memcpy(out_ids, sel.data(), sel.size() * sizeof(int32_t));
  if (out_evalues) memcpy(out_evalues, ev.data(), ev.size() * sizeof(double));
  return (int)sel.size();
}

int hhvr_prefilter_profile(const float* q_p, const float* pav, const double* lib, int32_t Lq, int32_t score_offset,
                           int32_t bit_factor, uint8_t* plain) {
  hhv::PrefilterQueryProfile(q_p, pav, lib, Lq, score_offset, bit_factor, plain);
  return 0;
}

int hhvr_read_context_library(const char* path, double* probs) {
  std::vector<double> v;
  std::string err;
  if (!hhv::ReadContextLibrary(path, &v, &err)) return -1;
  memcpy(probs, v.data(), v.size() * sizeof(double));
  return 219;
}

float hhvr_flog2(float x) { return hhv::flog2(x); }
float hhvr_fpow2(float x) { return hhv::fpow2(x); }

}  // extern "C"
