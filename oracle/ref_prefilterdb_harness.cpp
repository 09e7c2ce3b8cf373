// oracle/ref_prefilterdb_harness.cpp -- TEST INFRASTRUCTURE (never linked into the product).
//
// Constructs the reference's class Prefilter the way HHblitsDatabase::initPrefilter does (src/hhdatabase.cpp:128-131:
// new Prefilter(cs_library, cs219_database) on an FFindexDatabase of column-state sequences) and calls prefilter_db
// (src/hhdatabase.cpp:173-190).  Compiled twice into oracle/_ref/libhhref_dropin.so:
//   - as is:           ref_prefilterdb_run_cpu -> the reference's src/hhprefilter.cpp
//   - -DHARNESS_HIP:   ref_prefilterdb_run_hip -> hh-suite_amd/dropin/hhprefilter_hip.cpp (class renamed by macro)
#ifdef HARNESS_HIP
#define Prefilter PrefilterHip
#define RUN_NAME ref_prefilterdb_run_hip
#else
#define RUN_NAME ref_prefilterdb_run_cpu
#endif

#include <sys/time.h>

#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "ffindexdatabase.h"
#include "hhprefilter.h"

extern "C" {

// q_p: rows p[0..Lq-1][20] of the prefilter query, pav[20].  previous: n_prev template names (without extension) that
// count as searched before.  ipar: threads, gap_open, gap_extend, score_offset, bit_factor, smax_thresh, min_hits, maxnumdb;
// dpar: evalue_thresh, evalue_coarse_thresh.  reps / best_seconds: prefilter_db is called reps times, best wall time out.  Output: names joined with '\n' into new_out / old_out (cap bytes each) and the
// lengths; returns new count + (old count << 20), or a negative error.
int RUN_NAME(const char* ffdata, const char* ffindex, const float* q_p, const float* pav, int Lq, const int* ipar,
             const double* dpar, int n_prev, const char* const* previous, char* new_out, int* new_len, char* old_out, int* old_len,
             int cap, int reps, double* best_seconds) {
  Log::reporting_level() = WARNING;
  FFindexDatabase db(ffdata, ffindex, false);
  Prefilter* pf = new Prefilter(std::string(""), &db);
  HMM* q = new HMM(MAXSEQDIS, Lq + 2);
  q->L = Lq;
  for (int i = 0; i < Lq; ++i)
    for (int a = 0; a < 20; ++a) q->p[i][a] = q_p[i * 20 + a];
  for (int a = 0; a < 20; ++a) q->pav[a] = pav[a];
  Hit dummy;
  Hash<Hit>* previous_hits = new Hash<Hit>(1631, dummy);
  for (int k = 0; k < n_prev; ++k) {
    std::stringstream ss;
    ss << previous[k] << "__" << 1;
    previous_hits->Add((char*)ss.str().c_str(), dummy);
  }
  float R[20][20];
  memset(R, 0, sizeof(R));
  std::vector<std::pair<int, std::string> > nw, old;
  double best = 1e30;
  for (int r = 0; r < (reps < 1 ? 1 : reps); ++r) {  // timing: the prefilter_db call alone, best of reps
    nw.clear();
    old.clear();
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    pf->prefilter_db(q, previous_hits, ipar[0], ipar[1], ipar[2], ipar[3], ipar[4], dpar[0], dpar[1], ipar[5], ipar[6], ipar[7], R,
                     nw, old);
    gettimeofday(&t1, NULL);
    best = std::min(best, (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec));
  }
  if (best_seconds) *best_seconds = best;
  std::string a, b;
  for (size_t k = 0; k < nw.size(); ++k) {
    a += nw[k].second + "\n";
    new_len[k] = nw[k].first;
  }
  for (size_t k = 0; k < old.size(); ++k) {
    b += old[k].second + "\n";
    old_len[k] = old[k].first;
  }
  if ((int)a.size() >= cap || (int)b.size() >= cap) return -2;
  strcpy(new_out, a.c_str());
  strcpy(old_out, b.c_str());
  delete previous_hits;
  delete q;
  delete pf;
  return (int)nw.size() + ((int)old.size() << 20);
}

}  // extern "C"
