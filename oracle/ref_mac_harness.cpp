// oracle/ref_mac_harness.cpp -- TEST INFRASTRUCTURE: the reference's MAC realignment
//   PosteriorDecoder::realign            src/hhposteriordecoder.cpp:86-119
//     initializeForAlignment / maskViterbiAlignment / excludeMACAlignment   :151-262
//     forwardAlgorithm   src/hhforwardalgorithm.cpp:10-219
//     backwardAlgorithm  src/hhbackwardalgorithm.cpp:10-135
//     macAlgorithm       src/hhmacalgorithm.cpp:18-179
//     backtraceMAC       src/hhbacktracemac.cpp:113-272
//     writeProfilesToHits  src/hhbacktracemac.cpp:14-110 (the sparse lists of the -o_matrices output)
// run on two prepared HMMs given as tensors (the same p / log2-tr arrays the Viterbi harness takes) and a
// Viterbi hit (end points + path).  The harness does what PosteriorDecoderRunner::executeComputation does around
// realign (src/hhposteriordecoderrunner.cpp:43-119): Log2LinTransitionProbs(1.0) on both HMMs,
// initializeQueryHMMTransitions, and it hands back the linear transitions so that a test can feed the product the
// very same numbers.
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

#include "hhposteriordecoder.h"
#include "hhviterbimatrix.h"
#include "hhhitlist-inl.h"

namespace {
HMM* make_hmm(const float* p, const float* tr_log, int L) {
  HMM* h = new HMM(MAXSEQDIS, L + 2);
  h->L = L;
  for (int i = 0; i <= L; ++i) {
    for (int a = 0; a < 20; ++a) h->p[i][a] = p[(size_t)i * 20 + a];
    for (int a = 0; a < 7; ++a) h->tr[i][a] = tr_log[(size_t)i * 7 + a];
  }
  h->trans_lin = 0;
  h->nss_dssp = -1;
  h->nss_pred = -1;
  h->mu = 0;
  return h;
}
float zS73[NDSSP][NSSPRED][MAXCF];
float zS33[NSSPRED][MAXCF][NSSPRED][MAXCF];
float zS37[NSSPRED][MAXCF][NDSSP];
// what writeProfilesToHits (src/hhbacktracemac.cpp:14-110) attached to the hit of the last ref_mac_realign call:
// lists 0 forward, 1 backward, 2 posterior as (i, j, value), and the two profiles
std::vector<float> g_list[3], g_profile[2];
}  // namespace

extern "C" {

// n_prev previous MAC alignments of the same template (alt_i/alt_j lists, concatenated; prev_off[n_prev+1])
int ref_mac_realign(const float* q_p, const float* q_tr_log, int Lq, const float* t_p, const float* t_tr_log, int Lt,
                    int local, float shift, float mact, float corr, int par_min_overlap,
                    int vi1, int vj1, int vi2, int vj2, int v_nsteps, const int* v_i, const int* v_j,
                    int n_prev, const int* prev_off, const int* prev_i, const int* prev_j,
                    float* q_tr_lin, float* t_tr_lin,             // (L+1)*7 each, as the algorithms saw them
                    unsigned char* celloff,                        // (Lq+1)*(Lt+1): mask BEFORE forward
                    float* forward,                                // (Lq+1)*(Lt+1): p_mm after forward (scaled F_MM)
                    float* posterior,                              // (Lq+1)*(Lt+1): p_mm after backward
                    unsigned char* bmm,                            // (Lq+1)*(Lt+1): MAC backtrace codes after backtraceMAC
                    double* scale, double* Pforward,               // Lq+2 ; 1
                    int* o_scalars /* nsteps,i1,j1,i2,j2,matched_cols */, float* o_sum_of_probs,
                    int* o_i, int* o_j, char* o_states, float* o_S, float* o_P) {
  Log::reporting_level() = INFO;
  HMM* q = make_hmm(q_p, q_tr_log, Lq);
  HMM* t = make_hmm(t_p, t_tr_log, Lt);
  q->Log2LinTransitionProbs(1.0);
  t->Log2LinTransitionProbs(1.0);
  // PosteriorDecoderRunner::initializeQueryHMMTransitions (src/hhposteriordecoderrunner.cpp:146-155); the runner's
  // translation unit itself needs the database layer, so its eight assignments are restated here
  q->tr[0][M2D] = q->tr[0][M2I] = 0.0f;
  q->tr[0][I2M] = q->tr[0][I2I] = 0.0f;
  q->tr[0][D2M] = q->tr[0][D2D] = 0.0f;
  q->tr[Lq][M2M] = 1.0f;
  q->tr[Lq][M2D] = q->tr[Lq][M2I] = 0.0f;
  q->tr[Lq][I2M] = q->tr[Lq][I2I] = 0.0f;
  q->tr[Lq][D2M] = 1.0f;
  q->tr[Lq][D2D] = 0.0f;

  ViterbiMatrix vm;
  vm.AllocateBacktraceMatrix(Lq, Lt);
  PosteriorMatrix pm;
  pm.allocateMatrix(Lq, Lt);
  PosteriorDecoder dec(Lt, local != 0, Lq, 0.0f, zS73, zS33, zS37);

  Hit hit;
  hit.L = Lt;
  hit.self = 0;
  hit.ssm1 = hit.ssm2 = 0;
  hit.i1 = vi1;
  hit.j1 = vj1;
  hit.i2 = vi2;
  hit.j2 = vj2;
  hit.nsteps = v_nsteps;
  hit.i = new int[v_nsteps + 2];
  hit.j = new int[v_nsteps + 2];
  hit.states = new char[v_nsteps + 2];
  for (int s = 0; s <= v_nsteps; ++s) {
    hit.i[s] = v_i[s];
    hit.j[s] = v_j[s];
    hit.states[s] = 0;
  }
  hit.score = 1.0f;
  std::vector<std::vector<int> > pi(n_prev), pj(n_prev);
  std::vector<PosteriorDecoder::MACBacktraceResult> excl;
  for (int k = 0; k < n_prev; ++k) {
    pi[k].assign(prev_i + prev_off[k], prev_i + prev_off[k + 1]);
    pj[k].assign(prev_j + prev_off[k], prev_j + prev_off[k + 1]);
    excl.push_back(PosteriorDecoder::MACBacktraceResult(&pi[k], &pj[k]));
  }

  // realign(), split so that the mask can be observed before the DP runs (:92-119)
  dec.memorizeHitValues(hit);
  dec.initializeForAlignment(*q, *t, hit, vm, 0, t->L, par_min_overlap);
  for (size_t k = 0; k < excl.size(); ++k) dec.excludeMACAlignment(q->L, hit.L, vm, 0, excl[k]);
  for (int i = 0; i <= Lq; ++i)
    for (int j = 0; j <= Lt; ++j) celloff[(size_t)i * (Lt + 1) + j] = (i >= 1 && j >= 1 && vm.getCellOff(i, j, 0)) ? 1 : 0;
  for (int i = 0; i <= Lq; ++i)
    for (int a = 0; a < 7; ++a) q_tr_lin[(size_t)i * 7 + a] = q->tr[i][a];
  for (int j = 0; j <= Lt; ++j)
    for (int a = 0; a < 7; ++a) t_tr_lin[(size_t)j * 7 + a] = t->tr[j][a];
  dec.forwardAlgorithm(*q, *t, hit, pm, vm, shift, 0);
  *Pforward = hit.Pforward;
  for (int i = 0; i <= Lq + 1; ++i) scale[i] = dec.scale[i];
  for (int i = 0; i <= Lq; ++i)
    for (int j = 0; j <= Lt; ++j) forward[(size_t)i * (Lt + 1) + j] = pm.getPosteriorValue(i, j);
  dec.backwardAlgorithm(*q, *t, hit, pm, vm, shift, 0);
  for (int i = 0; i <= Lq; ++i)
    for (int j = 0; j <= Lt; ++j) posterior[(size_t)i * (Lt + 1) + j] = pm.getPosteriorValue(i, j);
  dec.macAlgorithm(*q, *t, hit, pm, vm, mact, 0);
  dec.backtraceMAC(*q, *t, pm, vm, 0, hit, corr);
  for (int i = 0; i <= Lq; ++i)
    for (int j = 0; j <= Lt; ++j) bmm[(size_t)i * (Lt + 1) + j] = (unsigned char)vm.getMatMat(i, j, 0);
  dec.writeProfilesToHits(*q, *t, pm, vm, hit);  // (realign() calls it last, :117)
  {
    float** const m[3] = {hit.forward_matrix, hit.backward_matrix, hit.posterior_matrix};
    const size_t n[3] = {hit.forward_entries, hit.backward_entries, hit.posterior_entries};
    for (int w = 0; w < 3; ++w) {
      g_list[w].clear();
      for (size_t e = 0; e < n[w]; ++e) g_list[w].insert(g_list[w].end(), m[w][e], m[w][e] + 3);
    }
    g_profile[0].assign(hit.forward_profile, hit.forward_profile + Lq + 1);
    g_profile[1].assign(hit.backward_profile, hit.backward_profile + Lq + 1);
  }
  o_scalars[0] = hit.nsteps;
  o_scalars[1] = hit.i1;
  o_scalars[2] = hit.j1;
  o_scalars[3] = hit.i2;
  o_scalars[4] = hit.j2;
  o_scalars[5] = hit.matched_cols;
  *o_sum_of_probs = hit.sum_of_probs;
  for (int s = 0; s <= hit.nsteps; ++s) {
    o_i[s] = hit.i[s];
    o_j[s] = hit.j[s];
    o_states[s] = hit.states[s];
    o_S[s] = s >= 1 ? hit.S[s] : 0.0f;
    o_P[s] = s >= 1 ? hit.P_posterior[s] : 0.0f;
  }
  delete q;
  delete t;
  return 0;
}

// CPU BASELINE of bench.py (next_rows.N4_mac_realign.reference_hits_per_s): n hits of one query realigned by the reference's
// member functions with the reference's parallelisation - PosteriorDecoderRunner::executeComputation's
//   #pragma omp parallel for schedule(static) num_threads(m_n_threads)      (src/hhposteriordecoderrunner.cpp:76)
// over the templates, one PosteriorDecoder + PosteriorMatrix + ViterbiMatrix per thread (:68, initializeConsumerThreads) -
// and the call sequence of PosteriorDecoder::realign (src/hhposteriordecoder.cpp:86-119) per hit.  Template k: columns
// col_off[k] .. of t_p ([.][20]) / t_tr_log ([.][7]), Viterbi path entries path_off[k] .. (entry 0 unused, 1 .. nsteps).
// Returns the seconds the parallel loop took (templates are built before it: the reference reads them inside its loop, but
// that is the database layer, not the realignment); *checksum = sum of the MAC alignments' step counts.
double ref_mac_realign_timed(const float* q_p, const float* q_tr_log, int Lq, int n, const long* col_off, const int* Lt,
                             const float* t_p, const float* t_tr_log, int local, float shift, float mact, float corr,
                             const int* ends /* [n][4] i1 j1 i2 j2 */, const int* nsteps, const long* path_off, const int* v_i,
                             const int* v_j, int threads, long* checksum) {
  Log::reporting_level() = INFO;
  HMM* q = make_hmm(q_p, q_tr_log, Lq);
  q->Log2LinTransitionProbs(1.0);
  q->tr[0][M2D] = q->tr[0][M2I] = 0.0f;
  q->tr[0][I2M] = q->tr[0][I2I] = 0.0f;
  q->tr[0][D2M] = q->tr[0][D2D] = 0.0f;
  q->tr[Lq][M2M] = 1.0f;
  q->tr[Lq][M2D] = q->tr[Lq][M2I] = 0.0f;
  q->tr[Lq][I2M] = q->tr[Lq][I2I] = 0.0f;
  q->tr[Lq][D2M] = 1.0f;
  q->tr[Lq][D2D] = 0.0f;
  int max_Lt = 1;
  for (int k = 0; k < n; ++k) max_Lt = max_Lt > Lt[k] ? max_Lt : Lt[k];
  std::vector<HMM*> ts(n);
  for (int k = 0; k < n; ++k) {
    ts[k] = make_hmm(t_p + (size_t)col_off[k] * 20, t_tr_log + (size_t)col_off[k] * 7, Lt[k]);
    ts[k]->Log2LinTransitionProbs(1.0);
  }
  if (threads < 1) threads = 1;
  std::vector<ViterbiMatrix*> vm(threads);
  std::vector<PosteriorMatrix*> pm(threads);
  std::vector<PosteriorDecoder*> dec(threads);
  for (int t = 0; t < threads; ++t) {
    vm[t] = new ViterbiMatrix();
    vm[t]->AllocateBacktraceMatrix(Lq, max_Lt);
    pm[t] = new PosteriorMatrix();
    pm[t]->allocateMatrix(Lq, max_Lt);
    dec[t] = new PosteriorDecoder(max_Lt, local != 0, Lq, 0.0f, zS73, zS33, zS37);
  }
  long sum = 0;
  const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : sum)
  for (int k = 0; k < n; ++k) {
    const int th = omp_get_thread_num();
    Hit hit;
    hit.L = Lt[k];
    hit.self = 0;
    hit.ssm1 = hit.ssm2 = 0;
    hit.i1 = ends[4 * k + 0];
    hit.j1 = ends[4 * k + 1];
    hit.i2 = ends[4 * k + 2];
    hit.j2 = ends[4 * k + 3];
    hit.nsteps = nsteps[k];
    hit.i = new int[nsteps[k] + 2];
    hit.j = new int[nsteps[k] + 2];
    hit.states = new char[nsteps[k] + 2];
    for (int s = 0; s <= nsteps[k]; ++s) {
      hit.i[s] = v_i[path_off[k] + s];
      hit.j[s] = v_j[path_off[k] + s];
      hit.states[s] = 0;
    }
    hit.score = 1.0f;
    HMM* t = ts[k];
    dec[th]->memorizeHitValues(hit);
    dec[th]->initializeForAlignment(*q, *t, hit, *vm[th], 0, t->L, 0);
    dec[th]->forwardAlgorithm(*q, *t, hit, *pm[th], *vm[th], shift, 0);
    dec[th]->backwardAlgorithm(*q, *t, hit, *pm[th], *vm[th], shift, 0);
    dec[th]->macAlgorithm(*q, *t, hit, *pm[th], *vm[th], mact, 0);
    dec[th]->backtraceMAC(*q, *t, *pm[th], *vm[th], 0, hit, corr);
    sum += hit.nsteps;
  }
  const double dt = omp_get_wtime() - t0;
  if (checksum) *checksum = sum;
  for (int t = 0; t < threads; ++t) {
    delete dec[t];
    delete pm[t];
    delete vm[t];
  }
  for (int k = 0; k < n; ++k) delete ts[k];
  delete q;
  return dt;
}

// The reference's sort key of a hit (tests/test_gpu_topk_pvalue.py): what HitList::CalculatePvalues (src/hhhitlist.cpp:499-531)
// leaves in Hit::score_aass - lamda_NN / mu_NN (src/hhhitlist-inl.h), logPvalue / Pvalue (src/hhhit-inl.h), CalcEvalScoreProbab
// (src/hhhit.h:134-141) - for n hits given by score, score_ss, template length and diversity.
int ref_score_aass(int n, const float* score, const float* score_ss, const int* Lt, const float* t_neff, int Lq, float q_neff, int loc,
                   int N_searched, float* out_score_aass, double* out_logPval) {
  const float log1000 = log(1000.0);
  for (int k = 0; k < n; ++k) {
    Hit hit;
    float lamda = LAMDA_GLOB, mu = 3.0;
    hit.score = score[k];
    hit.score_ss = score_ss[k];
    hit.L = Lt[k];
    hit.Neff_HMM = t_neff[k];
    hit.ssm1 = hit.ssm2 = 0;
    if (loc) {
      lamda = lamda_NN(log(Lq) / log1000, log(hit.L) / log1000, q_neff / 10.0, hit.Neff_HMM / 10.0);
      mu = mu_NN(log(Lq) / log1000, log(hit.L) / log1000, q_neff / 10.0, hit.Neff_HMM / 10.0);
    }
    hit.logPval = logPvalue(hit.score, lamda, mu);
    hit.Pval = Pvalue(hit.score, lamda, mu);
    hit.CalcEvalScoreProbab(N_searched, lamda, (char)loc, 0, 0.0f);
    out_score_aass[k] = hit.score_aass;
    if (out_logPval) out_logPval[k] = hit.logPval;
  }
  return 0;
}

// list `which` (0 forward, 1 backward, 2 posterior) of the last ref_mac_realign call: returns the number of entries and
// copies up to cap of them as float triples (i, j, value) - the layout of Hit::forward_matrix[e][0..2]
long ref_mac_last_list(int which, long cap, float* triples) {
  const std::vector<float>& l = g_list[which];
  const long n = (long)(l.size() / 3);
  for (long e = 0; e < n && e < cap; ++e)
    for (int c = 0; c < 3; ++c) triples[e * 3 + c] = l[(size_t)e * 3 + c];
  return n;
}
// Hit::forward_profile (0) / backward_profile (1) of the last call, Lq + 1 floats
int ref_mac_last_profile(int which, float* out) {
  for (size_t i = 0; i < g_profile[which].size(); ++i) out[i] = g_profile[which][i];
  return (int)g_profile[which].size();
}

}  // extern "C"
