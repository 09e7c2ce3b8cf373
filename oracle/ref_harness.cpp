// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" harness around the *unmodified* reference implementation of
// the Viterbi hot path.  It is compiled by oracle/Makefile together with the
// reference's own translation units, taken from where they lie under
// /root/reference (nothing is copied into this repository):
//
//   src/hhviterbialgorithm.cpp  (x4: plain, -DVITERBI_CELLOFF, -DVITERBI_SS_SCORE, both)
//   src/hhviterbi.cpp  src/hhhmmsimd.cpp  src/hhviterbimatrix.cpp
//
// with the reference's pinned flags (-mavx2, no FMA contraction) into
// oracle/_ref/libhhref.so.  The harness only *calls* the reference:
//   HMMSimd::MapHMMVector / MapOneHMM     (src/hhhmmsimd.cpp:73-160)
//   Viterbi::Align                        (src/hhviterbi.cpp:163-191)
//   Viterbi::Backtrace                    (src/hhviterbi.cpp:83-160)
//   Viterbi::ScoreForBacktrace            (src/hhviterbi.cpp:195-281)
//   Viterbi::ExcludeAlignment             (src/hhviterbi.cpp:61-77)
//   log2f4 / fast_log2 / ScalarProd20     (src/hhutil-inl.h:509, src/util-inl.h:108, src/hhhit-inl.h:61)
//
// The HMM class of the reference (src/hhhmm.h) drags in the whole text parser
// and pseudocount engine through its constructor.  The Viterbi path only reads
// the fields L, p, tr, ss_*, mu, lamda, so the harness builds zero-initialised
// HMM shells with calloc() (no constructor call, hence no link dependency on
// hhhmm.cpp) and fills exactly those fields.  This needs -fno-access-control
// because tr/ss_* are private (src/hhhmm.h:147-155).
//
// Input convention ("prepared tensors", i.e. what Viterbi::Align sees after
// PrepareTemplateHMM): per HMM of length L
//   p  : (L+1) x 20 floats, row 0 unused, p[i][a]                       (AoS)
//   tr : (L+1) x 7  floats in the reference enum order
//        M2M,M2I,M2D,I2M,I2I,D2M,D2D (src/hhdecl.h:68), log2 space
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <chrono>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "hhviterbi.h"
#include "hhviterbimatrix.h"
#include "hhhmmsimd.h"
#include "hhhit.h"

extern "C" void ref_init_fast_log2_like_hhsearch();  // ref_hmm_harness.cpp

namespace {

struct HmmShell {
  HMM* hmm;
  int cap;  // rows allocated
  std::vector<float*> prow, trow;
  float* pbuf;
  float* trbuf;
  char* ss;  // zero filled, shared for ss_pred/ss_conf/ss_dssp when no SS given
  char* ss_pred;
  char* ss_conf;
  char* ss_dssp;
};

static HmmShell* shell_new(int maxres) {
  HmmShell* s = new HmmShell();
  s->cap = maxres;
  s->hmm = (HMM*)calloc(1, sizeof(HMM));  // no ctor: see header comment
  s->pbuf = (float*)mem_align(ALIGN_FLOAT, (size_t)maxres * 32 * sizeof(float));   // 32-float pitch keeps rows 32B aligned
  s->trbuf = (float*)mem_align(ALIGN_FLOAT, (size_t)maxres * 8 * sizeof(float));
  memset(s->pbuf, 0, (size_t)maxres * 32 * sizeof(float));
  memset(s->trbuf, 0, (size_t)maxres * 8 * sizeof(float));
  s->prow.resize(maxres);
  s->trow.resize(maxres);
  for (int i = 0; i < maxres; i++) {
    s->prow[i] = s->pbuf + (size_t)i * 32;
    s->trow[i] = s->trbuf + (size_t)i * 8;
  }
  s->ss_pred = (char*)calloc(maxres + 1, 1);
  s->ss_conf = (char*)calloc(maxres + 1, 1);
  s->ss_dssp = (char*)calloc(maxres + 1, 1);
  s->hmm->p = s->prow.data();
  s->hmm->tr = s->trow.data();
  s->hmm->ss_pred = s->ss_pred;
  s->hmm->ss_conf = s->ss_conf;
  s->hmm->ss_dssp = s->ss_dssp;
  s->hmm->L = 0;
  s->hmm->mu = 0;
  s->hmm->lamda = 0;
  return s;
}

static void shell_free(HmmShell* s) {
  free(s->pbuf);
  free(s->trbuf);
  free(s->ss_pred);
  free(s->ss_conf);
  free(s->ss_dssp);
  free(s->hmm);
  delete s;
}

static void shell_fill(HmmShell* s, const float* p, const float* tr, int L,
                       const signed char* ss_pred, const signed char* ss_conf, const signed char* ss_dssp) {
  s->hmm->L = L;
  for (int i = 0; i <= L; i++) {
    memcpy(s->prow[i], p + (size_t)i * 20, 20 * sizeof(float));
    memcpy(s->trow[i], tr + (size_t)i * 7, 7 * sizeof(float));
  }
  memset(s->ss_pred, 0, s->cap + 1);
  memset(s->ss_conf, 0, s->cap + 1);
  memset(s->ss_dssp, 0, s->cap + 1);
  if (ss_pred) memcpy(s->ss_pred, ss_pred, L + 1);
  if (ss_conf) memcpy(s->ss_conf, ss_conf, L + 1);
  if (ss_dssp) memcpy(s->ss_dssp, ss_dssp, L + 1);
}

struct RefCtx {
  int maxres;
  int local;
  float egq, egt, corr, shift, ssw;
  int ss_mode;
  float S73[NDSSP][NSSPRED][MAXCF];
  float S33[NSSPRED][MAXCF][NSSPRED][MAXCF];
  float S37[NSSPRED][MAXCF][NDSSP];
  Viterbi* vit;
  ViterbiMatrix* mat;
  HMMSimd* qs;
  HMMSimd* ts;
  HmmShell* q;
  HmmShell* t[VECSIZE_FLOAT];
};

}  // namespace

extern "C" {

int ref_vecsize() { return VECSIZE_FLOAT; }

// ---- unit-level probes (pin the restated math functions) -------------------
float ref_log2f4(float x) {
  simd_float v = simdf32_set(x);
  simd_float r = log2f4(v);
  float out[VECSIZE_FLOAT] __attribute__((aligned(ALIGN_FLOAT)));
  simdf32_store(out, r);
  return out[0];
}
float ref_fast_log2(float x) {
  ref_init_fast_log2_like_hhsearch();
  return fast_log2(x);
}
// scalar (SSE-ordered) product used when re-scoring the backtrace; q,t must hold 20 floats
float ref_scalarprod20(const float* q, const float* t) {
  float qa[20] __attribute__((aligned(32)));
  float ta[20] __attribute__((aligned(32)));
  memcpy(qa, q, sizeof(qa));
  memcpy(ta, t, sizeof(ta));
  return ScalarProd20(qa, ta);
}
// lane 0 of the vector product used inside the DP
float ref_scalarprod20vec(const float* q, const float* t) {
  simd_float qv[20], tv[20];
  for (int a = 0; a < 20; a++) {
    qv[a] = simdf32_set(q[a]);
    tv[a] = simdf32_set(t[a]);
  }
  simd_float r = Viterbi::ScalarProd20Vec(qv, tv);
  float out[VECSIZE_FLOAT] __attribute__((aligned(ALIGN_FLOAT)));
  simdf32_store(out, r);
  return out[0];
}

// ---- context ---------------------------------------------------------------
void* ref_create(int maxres, int local, float egq, float egt, float corr, float shift, int ss_mode, float ssw,
                 const float* S73, const float* S33, const float* S37) {
  // The reference's default log level is DEBUG4 (src/log.h:58); every app lowers it from -v
  // (default 2 = INFO) before touching the Viterbi code.  At >= DEBUG1 ScoreForBacktrace would call
  // PrintDebug, which dereferences sequence arrays our HMM shells do not have.
  Log::reporting_level() = INFO;
  ref_init_fast_log2_like_hhsearch();
  RefCtx* c = new RefCtx();
  c->maxres = maxres;
  c->local = local;
  c->egq = egq;
  c->egt = egt;
  c->corr = corr;
  c->shift = shift;
  c->ssw = ssw;
  c->ss_mode = ss_mode;
  memset(c->S73, 0, sizeof(c->S73));
  memset(c->S33, 0, sizeof(c->S33));
  memset(c->S37, 0, sizeof(c->S37));
  if (S73) memcpy(c->S73, S73, sizeof(c->S73));
  if (S33) memcpy(c->S33, S33, sizeof(c->S33));
  if (S37) memcpy(c->S37, S37, sizeof(c->S37));
  c->vit = new Viterbi(maxres, local != 0, egq, egt, corr, 0, shift, ss_mode, ssw, c->S73, c->S33, c->S37);
  c->mat = new ViterbiMatrix();
  c->qs = new HMMSimd(maxres);
  c->ts = new HMMSimd(maxres);
  c->q = shell_new(maxres);
  for (int e = 0; e < VECSIZE_FLOAT; e++) c->t[e] = shell_new(maxres);
  return c;
}

void ref_destroy(void* h) {
  RefCtx* c = (RefCtx*)h;
  delete c->vit;
  delete c->mat;
  delete c->qs;
  delete c->ts;
  shell_free(c->q);
  for (int e = 0; e < VECSIZE_FLOAT; e++) shell_free(c->t[e]);
  delete c;
}

int ref_set_query(void* h, const float* p, const float* tr, int L, const signed char* ss_pred,
                  const signed char* ss_conf, const signed char* ss_dssp) {
  RefCtx* c = (RefCtx*)h;
  if (L + 2 > c->maxres) return -1;
  shell_fill(c->q, p, tr, L, ss_pred, ss_conf, ss_dssp);
  c->qs->MapOneHMM(c->q->hmm);
  return 0;
}

// One call of the reference batch unit: n <= VECSIZE_FLOAT templates mapped into the SIMD lanes
// exactly as ViterbiRunner does (src/hhviterbirunner.cpp:151-167), or n==1 && replicate!=0 ->
// MapOneHMM (the template replicated into all lanes = "single-length batch").
//
// celloff[e] (nullable, per lane nullable): (Lq+1) x (Lt_e+1) bytes, non-zero = cell excluded.
// Outputs (all per lane e < n):
//   score/i2/j2            ViterbiResult
//   bt[e]     (nullable)   (Lq+1) x (Ltb+1) bytes, Ltb = max L of the batch, copy of the byte matrix
//   With want_path != 0: Backtrace + ScoreForBacktrace:
//   nsteps, matched_cols, i_steps/j_steps/states/S (each cap entries per lane, index 0 unused),
//   hit_score (= BacktraceScore.score), score_ss
int ref_align_batch(void* h, int n, int replicate, const int* L, const float* const* p, const float* const* tr,
                    const signed char* const* ss_pred, const signed char* const* ss_conf,
                    const signed char* const* ss_dssp, int ss_hmm_mode, const unsigned char* const* celloff,
                    float* score, int* i2, int* j2, unsigned char* const* bt, int want_path, int cap, int* nsteps,
                    int* matched_cols, int* const* i_steps, int* const* j_steps, signed char* const* states,
                    float* const* S, float* hit_score, float* score_ss) {
  RefCtx* c = (RefCtx*)h;
  const int V = VECSIZE_FLOAT;
  if (n < 1 || n > V) return -1;
  const int Lq = c->q->hmm->L;
  int Ltb = 0;
  for (int e = 0; e < n; e++) {
    if (L[e] + 2 > c->maxres) return -2;
    Ltb = std::max(Ltb, L[e]);
    shell_fill(c->t[e], p[e], tr[e], L[e], ss_pred ? ss_pred[e] : NULL, ss_conf ? ss_conf[e] : NULL,
               ss_dssp ? ss_dssp[e] : NULL);
  }
  int lanes = n;
  if (n == 1 && replicate) {
    c->ts->MapOneHMM(c->t[0]->hmm);
    lanes = V;
  } else {
    std::vector<HMM*> v;
    for (int e = 0; e < n; e++) v.push_back(c->t[e]->hmm);
    c->ts->MapHMMVector(v);
  }
  c->mat->AllocateBacktraceMatrix(Lq, Ltb);
  // cell-off masks (src/hhviterbirunner.cpp:152-164 set them through ExcludeAlignment / setCellOff)
  c->mat->setCellOff(false);
  if (celloff) {
    for (int e = 0; e < n; e++) {
      if (!celloff[e]) continue;
      for (int i = 1; i <= Lq; i++)
        for (int j = 1; j <= L[e]; j++)
          if (celloff[e][(size_t)i * (L[e] + 1) + j]) {
            if (n == 1 && replicate) {
              for (int l = 0; l < V; l++) c->mat->setCellOff(i, j, l, true);
            } else {
              c->mat->setCellOff(i, j, e, true);
            }
          }
    }
  }
  Viterbi::ViterbiResult* r = c->vit->Align(c->qs, c->ts, c->mat, lanes, ss_hmm_mode);
  for (int e = 0; e < n; e++) {
    score[e] = r->score[e];
    i2[e] = r->i[e];
    j2[e] = r->j[e];
    if (bt && bt[e]) {
      for (int i = 0; i <= Lq; i++) {
        unsigned char* row = c->mat->getRow(i);
        for (int j = 0; j <= Ltb; j++) bt[e][(size_t)i * (Ltb + 1) + j] = row[j * V + e];
      }
    }
    if (want_path) {
      Viterbi::BacktraceResult b = Viterbi::Backtrace(c->mat, e, r->i, r->j);
      Viterbi::BacktraceScore bs = c->vit->ScoreForBacktrace(c->qs, c->ts, e, &b, r->score, ss_hmm_mode);
      nsteps[e] = b.count;
      matched_cols[e] = b.matched_cols;
      hit_score[e] = bs.score;
      score_ss[e] = bs.score_ss;
      for (int s = 1; s <= b.count && s < cap; s++) {
        i_steps[e][s] = b.i_steps[s];
        j_steps[e][s] = b.j_steps[s];
        states[e][s] = b.states[s];
        S[e][s] = bs.S[s];
      }
      delete[] b.i_steps;
      delete[] b.j_steps;
      delete[] b.states;
      delete[] bs.S;
      delete[] bs.S_ss;
    }
  }
  delete r;
  return Ltb;
}

// Viterbi::ExcludeAlignment on a fresh matrix; mask out: (Lq+1) x (Lt+1) bytes (1 = cell off).
int ref_exclude_alignment(int Lq, int Lt, const int* i_steps, const int* j_steps, int nsteps, unsigned char* mask) {
  const int maxres = std::max(Lq, Lt) + 2;
  HmmShell* q = shell_new(maxres);
  HmmShell* t = shell_new(maxres);
  q->hmm->L = Lq;
  t->hmm->L = Lt;
  HMMSimd qs(maxres), ts(maxres);
  qs.MapOneHMM(q->hmm);
  ts.MapOneHMM(t->hmm);
  ViterbiMatrix m;
  m.AllocateBacktraceMatrix(Lq, Lt);
  Viterbi::ExcludeAlignment(&m, &qs, &ts, 0, const_cast<int*>(i_steps), const_cast<int*>(j_steps), nsteps);
  for (int i = 0; i <= Lq; i++)
    for (int j = 0; j <= Lt; j++) mask[(size_t)i * (Lt + 1) + j] = (i >= 1 && j >= 1 && m.getCellOff(i, j, 0)) ? 1 : 0;
  shell_free(q);
  shell_free(t);
  return 0;
}

// ---- CPU baseline: the reference's own batch loop, timed --------------------
// Mirrors src/hhviterbirunner.cpp:117-168 for prepared profiles: templates are taken in the given
// order VECSIZE_FLOAT at a time, "#pragma omp parallel for schedule(dynamic,1)" over batches, each
// thread owning its Viterbi / HMMSimd / ViterbiMatrix.  Times MapHMMVector and Align separately
// (summed over threads is not meaningful; wall time of the whole loop is returned too).
// out: score/i2/j2 per template.  Returns wall seconds of the batch loop (map + align).
double ref_bench_align(int maxres, int local, float egq, float egt, float corr, float shift, const float* qp,
                       const float* qtr, int Lq, int N, const int* L, const float* const* p,
                       const float* const* tr, int threads, float* score, int* i2, int* j2, double* map_seconds) {
  const int V = VECSIZE_FLOAT;
  if (threads < 1) threads = 1;
  std::vector<RefCtx*> ctx(threads);
  int Lmax = 0;
  for (int k = 0; k < N; k++) Lmax = std::max(Lmax, L[k]);
  for (int t = 0; t < threads; t++) {
    ctx[t] = (RefCtx*)ref_create(maxres, local, egq, egt, corr, shift, 0, 0.0f, NULL, NULL, NULL);
    ref_set_query(ctx[t], qp, qtr, Lq, NULL, NULL, NULL);
    ctx[t]->mat->AllocateBacktraceMatrix(Lq, Lmax);
  }
  const int nb = (N + V - 1) / V;
  std::vector<double> mapsec(threads, 0.0);
  auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int b = 0; b < nb; b++) {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    RefCtx* c = ctx[tid];
    const int n = std::min(V, N - b * V);
    auto m0 = std::chrono::steady_clock::now();
    std::vector<HMM*> v;
    for (int e = 0; e < n; e++) {
      const int k = b * V + e;
      shell_fill(c->t[e], p[k], tr[k], L[k], NULL, NULL, NULL);
      v.push_back(c->t[e]->hmm);
    }
    c->ts->MapHMMVector(v);
    auto m1 = std::chrono::steady_clock::now();
    mapsec[tid] += std::chrono::duration<double>(m1 - m0).count();
    Viterbi::ViterbiResult* r = c->vit->Align(c->qs, c->ts, c->mat, n, 0);
    for (int e = 0; e < n; e++) {
      const int k = b * V + e;
      score[k] = r->score[e];
      i2[k] = r->i[e];
      j2[k] = r->j[e];
    }
    delete r;
  }
  auto t1 = std::chrono::steady_clock::now();
  double ms = 0;
  for (int t = 0; t < threads; t++) ms += mapsec[t];
  if (map_seconds) *map_seconds = ms;
  for (int t = 0; t < threads; t++) ref_destroy(ctx[t]);
  return std::chrono::duration<double>(t1 - t0).count();
}

// The reference's batch loop with backtrace: Align + Backtrace + ScoreForBacktrace for every lane of every batch
// (src/hhviterbirunner.cpp:12-71), OpenMP over batches like ViterbiRunner::alignment (:122).  replicate != 0 aligns every
// template alone (MapOneHMM = single-length batch, the parity definition for global mode over ragged lengths).
// Per template: ViterbiResult, alignment start, nsteps, matched_cols, Hit score and two checksums over the path
//   path_hash = sum_s ((i_s * 1000003 + j_s) * 31 + state_s) * (2 s + 1)   (mod 2^64, s = 1..nsteps)
//   s_hash    = sum_s bits(S_s) * (2 s + 1)
// which the GPU test recomputes from the engine's path pool.  Returns wall seconds.
double ref_bench_hits(int maxres, int local, float egq, float egt, float corr, float shift, const float* qp,
                      const float* qtr, int Lq, int N, const int* L, const float* const* p, const float* const* tr,
                      int threads, int replicate, float* score, int* i2, int* j2, int* i1, int* j1, int* nsteps,
                      int* matched_cols, float* hit_score, unsigned long long* path_hash, unsigned long long* s_hash) {
  const int V = replicate ? 1 : VECSIZE_FLOAT;
  if (threads < 1) threads = 1;
  std::vector<RefCtx*> ctx(threads);
  int Lmax = 0;
  for (int k = 0; k < N; k++) Lmax = std::max(Lmax, L[k]);
  for (int t = 0; t < threads; t++) {
    ctx[t] = (RefCtx*)ref_create(maxres, local, egq, egt, corr, shift, 0, 0.0f, NULL, NULL, NULL);
    ref_set_query(ctx[t], qp, qtr, Lq, NULL, NULL, NULL);
    ctx[t]->mat->AllocateBacktraceMatrix(Lq, Lmax);
  }
  const int nb = (N + V - 1) / V;
  auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int b = 0; b < nb; b++) {
    int tid = 0;
#ifdef _OPENMP
    tid = omp_get_thread_num();
#endif
    RefCtx* c = ctx[tid];
    const int n = std::min(V, N - b * V);
    std::vector<HMM*> v;
    for (int e = 0; e < n; e++) {
      const int k = b * V + e;
      shell_fill(c->t[e], p[k], tr[k], L[k], NULL, NULL, NULL);
      v.push_back(c->t[e]->hmm);
    }
    int lanes = n;
    if (replicate) {
      c->ts->MapOneHMM(c->t[0]->hmm);
      lanes = VECSIZE_FLOAT;
    } else {
      c->ts->MapHMMVector(v);
    }
    c->mat->setCellOff(false);
    Viterbi::ViterbiResult* r = c->vit->Align(c->qs, c->ts, c->mat, lanes, 0);
    for (int e = 0; e < n; e++) {
      const int k = b * V + e;
      score[k] = r->score[e];
      i2[k] = r->i[e];
      j2[k] = r->j[e];
      Viterbi::BacktraceResult bt = Viterbi::Backtrace(c->mat, e, r->i, r->j);
      Viterbi::BacktraceScore bs = c->vit->ScoreForBacktrace(c->qs, c->ts, e, &bt, r->score, 0);
      nsteps[k] = bt.count;
      matched_cols[k] = bt.matched_cols;
      hit_score[k] = bs.score;
      i1[k] = bt.i_steps[bt.count];
      j1[k] = bt.j_steps[bt.count];
      unsigned long long h = 0, hs = 0;
      for (int s = 1; s <= bt.count; s++) {
        const unsigned long long w = 2ull * (unsigned long long)s + 1ull;
        h += (((unsigned long long)bt.i_steps[s] * 1000003ull + (unsigned long long)bt.j_steps[s]) * 31ull +
              (unsigned long long)(unsigned char)bt.states[s]) * w;
        unsigned int bits;
        memcpy(&bits, &bs.S[s], 4);
        hs += (unsigned long long)bits * w;
      }
      path_hash[k] = h;
      s_hash[k] = hs;
      delete[] bt.i_steps;
      delete[] bt.j_steps;
      delete[] bt.states;
      delete[] bs.S;
      delete[] bs.S_ss;
    }
    delete r;
  }
  auto t1 = std::chrono::steady_clock::now();
  for (int t = 0; t < threads; t++) ref_destroy(ctx[t]);
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
