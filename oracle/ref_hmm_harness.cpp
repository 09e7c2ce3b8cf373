// oracle/ref_hmm_harness.cpp -- TEST INFRASTRUCTURE (fixture generation only, build container only).
//
// Turns a real .hhm file (the reference ships data/query.hhm) into *prepared* profiles with the
// reference's own code, so that the parity fixtures include genuine HH-suite profile columns and
// transition scores (not only synthetic ones).  Compiled by oracle/Makefile from the reference's
// own sources where they lie (src/hhhmm.cpp, hhutil.cpp, util.cpp, hhdecl.cpp, hhmatrices.cpp,
// src/cs/aa.cc) into oracle/_ref/libhhref_hmm.so; nothing is copied.
//
// It calls, in the order of PrepareQueryHMM / PrepareTemplateHMM (src/hhfunc.cpp:121-160,165-202,
// "-nocontxt" branch = substitution-matrix pseudocounts, the only mode that works without the
// context_data.crf blob listed in .MISSING_LARGE_BLOBS):
//   SetSubstitutionMatrix            src/hhmatrices.cpp:20-75
//   HMM::Read                        src/hhhmm.cpp:202-694
//   HMM::AddTransitionPseudocounts   src/hhhmm.cpp:1722-1806
//   HMM::PreparePseudocounts         src/hhhmm.cpp:1811-1815
//   HMM::AddAminoAcidPseudocounts    src/hhhmm.cpp:1874-1964
//   HMM::CalculateAminoAcidBackground src/hhhmm.cpp:1854-1868
//   HMM::IncludeNullModelInHMM       src/hhhmm.cpp:2059-2144   (template side)
#include <cstdio>
#include <cstring>

#include "hhdecl.h"
#include "hhhmm.h"
#include "hhmatrices.h"
#include "hhutil.h"

#include <omp.h>
extern "C" {

// fast_log2 (src/util-inl.h:108-130) keeps its lookup table in function-local statics that the FIRST caller
// in the process initialises - and the initialiser is compiled per translation unit: inside hhhmm.cpp
// `log(float(1024+i))` binds to double log(double), inside hhviterbi.cpp to logf.  Every real hhsearch /
// hhblits / hhalign run prepares the query with HMM::AddTransitionPseudocounts (src/hhfunc.cpp:129 ->
// src/hhhmm.cpp:1722-1806, which calls fast_log2) long before the Viterbi stage, so the table the Viterbi
// rescoring sees is hhhmm.cpp's.  This function makes the harness process behave the same way.
void ref_init_fast_log2_like_hhsearch() {
  static bool done = false;
  if (done) return;
  done = true;
  if (Log::reporting_level() > INFO) Log::reporting_level() = INFO;  // the apps lower the DEBUG4 default from -v
  HMM h(2, 8);
  h.L = 1;
  for (int i = 0; i < 8; ++i) {
    h.Neff_M[i] = h.Neff_I[i] = h.Neff_D[i] = 1.0f;
    for (int k = 0; k < 7; ++k) h.tr[i][k] = -1.0f;
  }
  h.AddTransitionPseudocounts(0.15f, 1.0f, 0.6f, 0.6f, 0.6f, 0.6f, 1.0f, 1.0f);
}

// role 0: prepare as query; role 1: prepare as template against the (already prepared) query file
// out_p: maxL+1 rows x 20, out_tr: maxL+1 rows x 7 (enum order).  Returns L, or a negative error.
int ref_prepare_hhm(const char* query_path, const char* template_path, int maxres, float* q_p, float* q_tr,
                    float* t_p, float* t_tr, int* Lq, int* Lt) {
  Log::reporting_level() = WARNING;
  Parameters par(0, NULL);
  par.maxres = maxres;
  par.nocontxt = 1;
  float pb[21];
  float P[20][20], R[20][20], S[20][20], Sim[20][20];
  SetSubstitutionMatrix(par.matrix, pb, P, R, S, Sim);

  HMM* q = new HMM(MAXSEQDIS, maxres);
  FILE* f = fopen(query_path, "r");
  if (!f) return -1;
  char path[NAMELEN] = "";
  if (!q->Read(f, par.maxcol, par.nseqdis, pb, path)) {
    fclose(f);
    return -2;
  }
  fclose(f);
  // PrepareQueryHMM, input_format == 0, par.nocontxt
  q->AddTransitionPseudocounts(par.gapd, par.gape, par.gapf, par.gapg, par.gaph, par.gapi, par.gapb, par.gapb);
  q->PreparePseudocounts(R);
  q->AddAminoAcidPseudocounts(par.pc_hhm_nocontext_mode, par.pc_hhm_nocontext_a, par.pc_hhm_nocontext_b,
                              par.pc_hhm_nocontext_c);
  q->CalculateAminoAcidBackground(pb);
  *Lq = q->L;
  for (int i = 0; i <= q->L; ++i) {
    for (int a = 0; a < 20; ++a) q_p[i * 20 + a] = (i == 0) ? 0.0f : q->p[i][a];
    for (int k = 0; k < 7; ++k) q_tr[i * 7 + k] = q->tr[i][k];
  }

  HMM* t = new HMM(MAXSEQDIS, maxres);
  f = fopen(template_path, "r");
  if (!f) return -3;
  if (!t->Read(f, par.maxcol, par.nseqdis, pb, path)) {
    fclose(f);
    return -4;
  }
  fclose(f);
  // PrepareTemplateHMM, format == 0
  t->AddTransitionPseudocounts(par.gapd, par.gape, par.gapf, par.gapg, par.gaph, par.gapi, par.gapb, par.gapb);
  t->PreparePseudocounts(R);
  t->AddAminoAcidPseudocounts(par.pc_hhm_nocontext_mode, par.pc_hhm_nocontext_a, par.pc_hhm_nocontext_b,
                              par.pc_hhm_nocontext_c);
  t->CalculateAminoAcidBackground(pb);
  t->IncludeNullModelInHMM(q, t, par.columnscore, par.half_window_size_local_aa_bg_freqs, pb);
  *Lt = t->L;
  for (int i = 0; i <= t->L; ++i) {
    for (int a = 0; a < 20; ++a) t_p[i * 20 + a] = (i == 0) ? 0.0f : t->p[i][a];
    for (int k = 0; k < 7; ++k) t_tr[i * 7 + k] = t->tr[i][k];
  }
  delete q;
  delete t;
  return 0;
}

// ---- raw (unprepared) HMMs: the inputs of PrepareQueryHMM / PrepareTemplateHMM ------------------------
// Raw layout: f[(L+2)*20] (rows 0..L+1), tr[(L+1)*7] raw log2 transitions as HMM::Read leaves them,
// neff[(L+1)*3] = Neff_M, Neff_I, Neff_D per column, Neff_HMM.
// pb_out[20]: the background AFTER the read - HMM::Read overwrites the caller's pb with the NULL line of the
// file (src/hhhmm.cpp:536-546), and that is the pb the subsequent preparation of this HMM uses in the reference.
int ref_read_hhm_raw(const char* path, int maxres, float* f, float* tr, float* neff, float* Neff_HMM, int* L,
                     float* pb_out) {
  if (Log::reporting_level() > WARNING) Log::reporting_level() = WARNING;
  Parameters par(0, NULL);
  par.maxres = maxres;
  float pb[21];
  float P[20][20], R[20][20], S[20][20], Sim[20][20];
  SetSubstitutionMatrix(par.matrix, pb, P, R, S, Sim);
  HMM* h = new HMM(MAXSEQDIS, maxres);
  FILE* fp = fopen(path, "r");
  if (!fp) return -1;
  char pth[NAMELEN] = "";
  if (!h->Read(fp, par.maxcol, par.nseqdis, pb, pth)) {
    fclose(fp);
    return -2;
  }
  fclose(fp);
  *L = h->L;
  *Neff_HMM = h->Neff_HMM;
  if (pb_out) memcpy(pb_out, pb, 20 * sizeof(float));
  for (int i = 0; i <= h->L + 1; ++i)
    for (int a = 0; a < 20; ++a) f[i * 20 + a] = h->f[i][a];
  for (int i = 0; i <= h->L; ++i) {
    for (int k = 0; k < 7; ++k) tr[i * 7 + k] = h->tr[i][k];
    neff[i * 3 + 0] = h->Neff_M[i];
    neff[i * 3 + 1] = h->Neff_I[i];
    neff[i * 3 + 2] = h->Neff_D[i];
  }
  delete h;
  return 0;
}

float ref_fpow2(float x) { return fpow2(x); }

// the substitution matrix side products the preparation needs (Gonnet default): pb[20], R[20][20]
int ref_substitution_matrix(float* pb_out, float* R_out) {
  Parameters par(0, NULL);
  float pb[21];
  float P[20][20], R[20][20], S[20][20], Sim[20][20];
  SetSubstitutionMatrix(par.matrix, pb, P, R, S, Sim);
  memcpy(pb_out, pb, 20 * sizeof(float));
  memcpy(R_out, R, 400 * sizeof(float));
  return 0;
}

// role 0: PrepareQueryHMM (src/hhfunc.cpp:121-160, input_format 0, -nocontxt); role 1: PrepareTemplateHMM
// (src/hhfunc.cpp:165-202, format 0) against a query whose average composition is q_pav[20].
// gap = {gapd, gape, gapf, gapg, gaph, gapi, gapb}; pc = {pcm, pca, pcb, pcc}.
// out_p[(L+2)*20] (rows 0 and L+1 = background, as the reference leaves them), out_tr[(L+1)*7], out_pav[20].
// pb_in: nullable override of the background (e.g. the NULL line an .hhm read left behind)
int ref_prepare_raw(int role, int L, const float* f, const float* tr, const float* neff, float Neff_HMM,
                    const float* q_pav, const float* gap, const float* pc, int columnscore, const float* pb_in,
                    float* out_p, float* out_tr, float* out_pav) {
  if (Log::reporting_level() > WARNING) Log::reporting_level() = WARNING;
  Parameters par(0, NULL);
  float pb[21];
  float P[20][20], R[20][20], S[20][20], Sim[20][20];
  SetSubstitutionMatrix(par.matrix, pb, P, R, S, Sim);
  if (pb_in) memcpy(pb, pb_in, 20 * sizeof(float));
  const int maxres = L + 3;
  HMM* h = new HMM(MAXSEQDIS, maxres);
  h->L = L;
  h->Neff_HMM = Neff_HMM;
  for (int i = 0; i <= L + 1; ++i)
    for (int a = 0; a < 20; ++a) h->f[i][a] = f[i * 20 + a];
  for (int i = 0; i <= L; ++i) {
    for (int k = 0; k < 7; ++k) h->tr[i][k] = tr[i * 7 + k];
    h->Neff_M[i] = neff[i * 3 + 0];
    h->Neff_I[i] = neff[i * 3 + 1];
    h->Neff_D[i] = neff[i * 3 + 2];
  }
  h->AddTransitionPseudocounts(gap[0], gap[1], gap[2], gap[3], gap[4], gap[5], gap[6], gap[6]);
  h->PreparePseudocounts(R);
  h->AddAminoAcidPseudocounts((char)pc[0], pc[1], pc[2], pc[3]);
  h->CalculateAminoAcidBackground(pb);
  if (role == 1) {
    HMM* q = new HMM(MAXSEQDIS, 8);
    for (int a = 0; a < 20; ++a) q->pav[a] = q_pav[a];
    h->IncludeNullModelInHMM(q, h, columnscore, par.half_window_size_local_aa_bg_freqs, pb);
    delete q;
  }
  for (int i = 0; i <= L + 1; ++i)
    for (int a = 0; a < 20; ++a) out_p[i * 20 + a] = h->p[i][a];
  for (int i = 0; i <= L; ++i)
    for (int k = 0; k < 7; ++k) out_tr[i * 7 + k] = h->tr[i][k];
  for (int a = 0; a < 20; ++a) out_pav[a] = h->pav[a];
  delete h;
  return 0;
}

// CPU BASELINE of bench.py (next_rows.N2_prepare.reference_*): PrepareTemplateHMM's call sequence (role 1 above) on n raw
// templates of one length L, repeated `reps` times, on `threads` OpenMP threads (the reference prepares a template inside the
// per-thread loops of its Viterbi / realignment runners, src/hhviterbirunner.cpp:117-140): seconds of the loop.
// f[n][(L+2)*20], tr[n][(L+1)*7], neff[n][(L+1)*3], neff_hmm[n]
double ref_prepare_raw_timed(int n, int reps, int L, const float* f, const float* tr, const float* neff, const float* neff_hmm,
                             const float* q_pav, const float* gap, const float* pc, int columnscore, const float* pb_in, int threads,
                             double* checksum) {
  if (Log::reporting_level() > WARNING) Log::reporting_level() = WARNING;
  Parameters par(0, NULL);
  float pb[21];
  float P[20][20], R[20][20], S[20][20], Sim[20][20];
  SetSubstitutionMatrix(par.matrix, pb, P, R, S, Sim);
  if (pb_in) memcpy(pb, pb_in, 20 * sizeof(float));
  if (threads < 1) threads = 1;
  fast_log2(1.0f);  // the table's first-caller initialisation outside the parallel region
  double sum = 0.0;
  const double t0 = omp_get_wtime();
#pragma omp parallel num_threads(threads) reduction(+ : sum)
  {
    HMM* h = new HMM(MAXSEQDIS, L + 3);
    HMM* q = new HMM(MAXSEQDIS, 8);
    for (int a = 0; a < 20; ++a) q->pav[a] = q_pav[a];
#pragma omp for schedule(static)
    for (int e = 0; e < n * reps; ++e) {
      const int k = e % n;
      const float* fk = f + (size_t)k * (L + 2) * 20;
      const float* trk = tr + (size_t)k * (L + 1) * 7;
      const float* nk = neff + (size_t)k * (L + 1) * 3;
      h->L = L;
      h->Neff_HMM = neff_hmm[k];
      h->trans_lin = 0;
      for (int i = 0; i <= L + 1; ++i)
        for (int a = 0; a < 20; ++a) h->f[i][a] = fk[i * 20 + a];
      for (int i = 0; i <= L; ++i) {
        for (int t = 0; t < 7; ++t) h->tr[i][t] = trk[i * 7 + t];
        h->Neff_M[i] = nk[i * 3 + 0];
        h->Neff_I[i] = nk[i * 3 + 1];
        h->Neff_D[i] = nk[i * 3 + 2];
      }
      h->AddTransitionPseudocounts(gap[0], gap[1], gap[2], gap[3], gap[4], gap[5], gap[6], gap[6]);
      h->PreparePseudocounts(R);
      h->AddAminoAcidPseudocounts((char)pc[0], pc[1], pc[2], pc[3]);
      h->CalculateAminoAcidBackground(pb);
      h->IncludeNullModelInHMM(q, h, columnscore, par.half_window_size_local_aa_bg_freqs, pb);
      sum += h->p[1][0] + h->tr[L][0];
    }
    delete q;
    delete h;
  }
  const double dt = omp_get_wtime() - t0;
  if (checksum) *checksum = sum;
  return dt;
}

}  // extern "C"
