// oracle/ref_prefilter_harness.cpp -- TEST INFRASTRUCTURE: the reference's two prefilter kernels
//   Prefilter::ungapped_sse_score   src/hhprefilter.cpp:214-278   (gapless profile/sequence score, uint8 SIMD)
//   Prefilter::swStripedByte        src/hhprefilter.cpp:70-212    (striped Smith-Waterman, uint8 SIMD, Farrar/Zhao)
// called on a plain [220][Lq] byte profile (219 column states + the ANY state): the harness stripes the
// profile exactly like Prefilter::stripe_query_profile does (src/hhprefilter.cpp:386-425) and calls the
// private member functions (they use no member data; -fno-access-control, calloc'ed object shell).
// hhprefilter.cpp includes the generated resource header cs219.lib.h; oracle/Makefile generates it from
// $(REF)/data/cs219.lib (the same bytes the reference's cmake ResourceCompiler embeds).
#include <cstdlib>
#include <cstring>

#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "hhprefilter.h"
extern unsigned char cs219_lib[];   // generated resource header, defined in hhprefilter.cpp's translation unit
extern unsigned int cs219_lib_len;

namespace {
unsigned char* stripe(const unsigned char* plain, int Lq, int offset, int* W_out) {
  const int element_count = VECSIZE_INT * 4;
  const int W = (Lq + element_count - 1) / element_count;
  unsigned char* qc = (unsigned char*)malloc_simd_int((size_t)220 * (Lq + element_count));
  for (int a = 0; a < 220; ++a) {
    int h = a * W * element_count;
    for (int i = 0; i < W; ++i) {
      int j = i;
      for (int k = 0; k < element_count; ++k) {
        qc[h++] = (j >= Lq) ? (unsigned char)offset : plain[(size_t)a * Lq + j];
        j += W;
      }
    }
  }
  *W_out = W;
  return qc;
}
}  // namespace

#include <omp.h>
#include <vector>
extern "C" {

int ref_prefilter_vecbytes() { return VECSIZE_INT * 4; }

// plain[220][Lq] (state-major), seqs: n_db sequences concatenated (values 0..219), offsets[n_db+1]
int ref_prefilter_scores(const unsigned char* plain, int Lq, const unsigned char* seqs, const long* offsets, int n_db,
                         int score_offset, int gap_init, int gap_extend, int* ungapped, int* gapped) {
  int W;
  unsigned char* qc = stripe(plain, Lq, score_offset, &W);
  const int element_count = VECSIZE_INT * 4;
  simd_int* ws = (simd_int*)malloc_simd_int(3 * (Lq + element_count));
  Prefilter* pf = (Prefilter*)calloc(1, sizeof(Prefilter));
  for (int n = 0; n < n_db; ++n) {
    unsigned char* s = const_cast<unsigned char*>(seqs + offsets[n]);
    const int len = (int)(offsets[n + 1] - offsets[n]);
    if (ungapped) ungapped[n] = pf->ungapped_sse_score(qc, Lq, s, len, (unsigned char)score_offset, ws);
    if (gapped) gapped[n] = pf->swStripedByte(qc, Lq, s, len, gap_init, gap_extend, ws, ws + W, ws + 2 * W, score_offset);
  }
  free(pf);
  free(ws);
  free(qc);
  return 0;
}

// CPU BASELINE of bench.py (next_rows.N3_prefilter.reference_*_cells_per_s): the reference's two kernels over the database with
// the reference's parallelisation (src/hhprefilter.cpp:466-479 and :528-556: #pragma omp parallel for schedule(static), one
// workspace per thread).  which: 0 = ungapped_sse_score, 1 = swStripedByte.  Returns the seconds of the loop.
double ref_prefilter_scores_timed(const unsigned char* plain, int Lq, const unsigned char* seqs, const long* offsets, int n_db,
                                  int score_offset, int gap_init, int gap_extend, int which, int threads, long* checksum) {
  int W;
  unsigned char* qc = stripe(plain, Lq, score_offset, &W);
  const int element_count = VECSIZE_INT * 4;
  if (threads < 1) threads = 1;
  std::vector<simd_int*> ws(threads);
  for (int t = 0; t < threads; ++t) ws[t] = (simd_int*)malloc_simd_int(3 * (Lq + element_count));
  Prefilter* pf = (Prefilter*)calloc(1, sizeof(Prefilter));
  long sum = 0;
  const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : sum)
  for (int n = 0; n < n_db; ++n) {
    simd_int* w = ws[omp_get_thread_num()];
    unsigned char* s = const_cast<unsigned char*>(seqs + offsets[n]);
    const int len = (int)(offsets[n + 1] - offsets[n]);
    sum += which == 0 ? pf->ungapped_sse_score(qc, Lq, s, len, (unsigned char)score_offset, w)
                      : pf->swStripedByte(qc, Lq, s, len, gap_init, gap_extend, w, w + W, w + 2 * W, score_offset);
  }
  const double dt = omp_get_wtime() - t0;
  if (checksum) *checksum = sum;
  for (int t = 0; t < threads; ++t) free(ws[t]);
  free(pf);
  free(qc);
  return dt;
}

float ref_flog2(float x) { return flog2(x); }

static cs::ContextLibrary<cs::AA>* load_cs219() {
  // the two lines of Prefilter::Prefilter (src/hhprefilter.cpp:32-44) for the built-in library
  FILE* fin = fmemopen((void*)cs219_lib, cs219_lib_len, "r");
  cs::ContextLibrary<cs::AA>* lib = new cs::ContextLibrary<cs::AA>(fin);
  fclose(fin);
  cs::TransformToLin(*lib);
  return lib;
}

// central-column probabilities of the 219 context states, as stripe_query_profile reads them (:367)
int ref_cs219_probs(float* out /* [219][20] */) {
  cs::ContextLibrary<cs::AA>* lib = load_cs219();
  for (int k = 0; k < (int)cs::AS219::kSize; ++k)
    for (int a = 0; a < 20; ++a) out[k * 20 + a] = (*lib)[k].probs[0][a];
  delete lib;
  return (int)cs::AS219::kSize;
}

int ref_cs219_probs_f64(double* out /* [219][20] */) {
  cs::ContextLibrary<cs::AA>* lib = load_cs219();
  for (int k = 0; k < (int)cs::AS219::kSize; ++k)
    for (int a = 0; a < 20; ++a) out[k * 20 + a] = (*lib)[k].probs[0][a];
  delete lib;
  return (int)cs::AS219::kSize;
}

// Prefilter::stripe_query_profile (:355-425) on a query given as p[i][a], i = 0..Lq-1 (the rows it reads), and pav;
// returns the profile de-striped to plain [220][Lq]
int ref_prefilter_profile(const float* q_p, const float* pav, int Lq, int score_offset, int bit_factor, unsigned char* plain) {
  Prefilter* pf = (Prefilter*)calloc(1, sizeof(Prefilter));
  pf->cs_lib = load_cs219();
  HMM* q = new HMM(MAXSEQDIS, Lq + 2);
  q->L = Lq;
  for (int i = 0; i < Lq; ++i)
    for (int a = 0; a < 20; ++a) q->p[i][a] = q_p[i * 20 + a];
  for (int a = 0; a < 20; ++a) q->pav[a] = pav[a];
  const int ec = VECSIZE_INT * 4, W = (Lq + ec - 1) / ec;
  unsigned char* qc = (unsigned char*)malloc_simd_int((size_t)220 * (Lq + ec));
  pf->stripe_query_profile(q, score_offset, bit_factor, W, qc);
  for (int a = 0; a < 220; ++a)
    for (int i = 0; i < W; ++i)
      for (int k = 0; k < ec; ++k) {
        const int j = k * W + i;
        if (j < Lq) plain[(size_t)a * Lq + j] = qc[(size_t)a * W * ec + i * ec + k];
      }
  free(qc);
  delete q;
  delete pf->cs_lib;
  free(pf);
  return 0;
}

// the whole of Prefilter::prefilter_db (:428-596) on an in-memory database: sequence n is named "<n>"; returns the
// selected sequence ids in the order the reference emits them (new_prefilter_hits; previous_hits is empty)
int ref_prefilter_db(const float* q_p, const float* pav, int Lq, const unsigned char* seqs, const long* offsets, int n_db,
                     int threads, int gap_open, int gap_extend, int score_offset, int bit_factor, double evalue_thresh,
                     double evalue_coarse_thresh, int smax_thresh, int min_hits, int maxnumdb, int* out_ids, int out_cap) {
  Prefilter* pf = (Prefilter*)calloc(1, sizeof(Prefilter));
  pf->cs_lib = load_cs219();
  pf->num_dbs = n_db;
  pf->first = (unsigned char**)malloc(sizeof(unsigned char*) * n_db);
  pf->length = (int*)malloc(sizeof(int) * n_db);
  pf->dbnames = (char**)malloc(sizeof(char*) * n_db);
  for (int n = 0; n < n_db; ++n) {
    pf->first[n] = const_cast<unsigned char*>(seqs + offsets[n]);
    pf->length[n] = (int)(offsets[n + 1] - offsets[n]);
    pf->dbnames[n] = new char[16];
    snprintf(pf->dbnames[n], 16, "%d", n);
  }
  HMM* q = new HMM(MAXSEQDIS, Lq + 2);
  q->L = Lq;
  for (int i = 0; i < Lq; ++i)
    for (int a = 0; a < 20; ++a) q->p[i][a] = q_p[i * 20 + a];
  for (int a = 0; a < 20; ++a) q->pav[a] = pav[a];
  Hash<Hit>* previous = new Hash<Hit>(1631, Hit());
  float R[20][20];
  memset(R, 0, sizeof(R));
  std::vector<std::pair<int, std::string> > new_hits, old_hits;
  pf->prefilter_db(q, previous, threads, gap_open, gap_extend, score_offset, bit_factor, evalue_thresh, evalue_coarse_thresh,
                   smax_thresh, min_hits, maxnumdb, R, new_hits, old_hits);
  int n_out = 0;
  for (size_t k = 0; k < new_hits.size() && n_out < out_cap; ++k) out_ids[n_out++] = atoi(new_hits[k].second.c_str());
  delete previous;
  delete q;
  for (int n = 0; n < n_db; ++n) delete[] pf->dbnames[n];
  free(pf->dbnames);
  free(pf->length);
  free(pf->first);
  delete pf->cs_lib;
  free(pf);
  return n_out;
}

}  // extern "C"
