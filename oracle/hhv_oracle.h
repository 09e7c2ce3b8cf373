/* oracle/hhv_oracle.h -- TEST INFRASTRUCTURE (CPU restatement of the reference algorithm).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  The product (hh-suite_amd/) never includes, links or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the reference
 * itself (oracle/_ref/libhhref.so, built from /root/reference by oracle/Makefile) in
 * tests/test_oracle_vs_reference.py, and against the committed fixtures in tests/golden/ that
 * were generated from that reference build (tests/golden/make_golden.py).
 *
 * Data convention ("prepared tensors" = what Viterbi::Align sees after PrepareTemplateHMM):
 *   p  : (L+1) x 20 floats, row 0 unused
 *   tr : (L+1) x 7 floats, reference enum order M2M,M2I,M2D,I2M,I2I,D2M,D2D (src/hhdecl.h:68)
 */
#ifndef HHV_ORACLE_H
#define HHV_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

enum { HHO_M2M = 0, HHO_M2I = 1, HHO_M2D = 2, HHO_I2M = 3, HHO_I2I = 4, HHO_D2M = 5, HHO_D2D = 6 };
enum { HHO_STOP = 0, HHO_MM = 2, HHO_GD = 3, HHO_IM = 4, HHO_DG = 5, HHO_MI = 6 };
enum { HHO_NO_SS = 0, HHO_PRED_DSSP = 1, HHO_DSSP_PRED = 2, HHO_PRED_PRED = 4 };
enum { HHO_NDSSP = 8, HHO_NSSPRED = 4, HHO_MAXCF = 11 };

typedef struct {
  int local;   /* par.loc */
  float egq;   /* penalty_gap_query */
  float egt;   /* penalty_gap_template */
  float shift; /* par.shift */
  float corr;  /* par.corr */
  float ssw;   /* par.ssw */
  int ss_mode; /* par.ssm; 2 == Hit::SCORE_ALIGNMENT */
} hho_params;

/* secondary-structure inputs (all nullable when ss_hmm_mode == HHO_NO_SS) */
typedef struct {
  int ss_hmm_mode;
  const signed char *q_ss_pred, *q_ss_conf, *q_ss_dssp; /* [Lq+1] */
  const signed char *t_ss_pred, *t_ss_conf, *t_ss_dssp; /* [Lt+1] */
  const float *S73, *S33, *S37;                         /* [8][4][11], [4][11][4][11], [4][11][8] */
} hho_ss;

float hho_log2f4(float x);
float hho_fast_log2(float x);
float hho_dot20_vec(const float *q, const float *t); /* Viterbi::ScalarProd20Vec order */
/* switch: 0 (default) = the reference's arithmetic; 1 = study (tools/mfma_emission_study.py): emission dot product as one fmaf
 * chain, what an f32 MFMA would compute; 2 = the arithmetic of the engine's opt-in fused build (libhhviterbi_hip_fma.so).
 * Process-wide; no parity test of the default engine touches it. */
void hho_set_emission_mode(int mode);
float hho_dot20_scalar(const float *q, const float *t); /* ScalarProd20 order (plain C branch) */

/* One pair.  Lbatch >= Lt emulates the SIMD batch of the reference: columns Lt+1..Lbatch are the
 * padding MapHMMVector writes (p=0, tr=-FLT_MAX) and the global-mode "last column" is Lbatch.
 * Lbatch == Lt is the single-length batch (MapOneHMM).
 * celloff: nullable, (Lq+1) x (Lt+1) bytes, non-zero = excluded cell.
 * bt: nullable, (Lq+1) x (Lbatch+1) bytes, receives the backtrace bytes (rows/cols 0 zeroed). */
int hho_align(const hho_params *par, const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr,
              int Lt, int Lbatch, const unsigned char *celloff, const hho_ss *ss, float *score, int *i2, int *j2,
              unsigned char *bt);

/* Viterbi::Backtrace.  bt has row pitch `pitch`.  Arrays are 1-based with `cap` entries. */
int hho_backtrace(const unsigned char *bt, int pitch, int i2, int j2, int *i_steps, int *j_steps, signed char *states,
                  int cap, int *nsteps, int *matched_cols);

/* Viterbi::ScoreForBacktrace.  S: cap >= nsteps+1 floats. */
int hho_score_for_backtrace(const hho_params *par, const float *qp, const float *tp, const hho_ss *ss,
                            const int *i_steps, const int *j_steps, const signed char *states, int nsteps,
                            float viterbi_score, float *S, float *hit_score, float *score_ss);

/* Viterbi::ExcludeAlignment: ORs the +-40 cross mask of one path into mask ((Lq+1) x (Lt+1)). */
int hho_exclude_alignment(int Lq, int Lt, const int *i_steps, const int *j_steps, int nsteps, unsigned char *mask);

/* fpow2 (src/util-inl.h:190-215) */
float hho_fpow2(float x);

/* PrepareQueryHMM (role 0) / PrepareTemplateHMM (role 1), src/hhfunc.cpp:121-160,165-202 with input format 0 and
 * substitution-matrix pseudocounts (-nocontxt): AddTransitionPseudocounts, PreparePseudocounts,
 * AddAminoAcidPseudocounts, CalculateAminoAcidBackground and - templates - IncludeNullModelInHMM
 * (src/hhhmm.cpp:1722-1806,1811-1815,1874-1964,1854-1868,2059-2144).
 * Raw inputs: f[(L+2)*20], tr[(L+1)*7] (as HMM::Read leaves them), neff[(L+1)*3] = Neff_M,Neff_I,Neff_D, Neff_HMM;
 * pb[20], R[20*20] from SetSubstitutionMatrix; gap = {gapd,gape,gapf,gapg,gaph,gapi,gapb}; pc = {pcm,pca,pcb,pcc};
 * q_pav[20] = average composition of the prepared query (role 1).
 * Outputs: p[(L+2)*20], tr_out[(L+1)*7], pav[20]. */
int hho_prepare(int role, int L, const float *f, const float *tr, const float *neff, float Neff_HMM, const float *pb,
                const float *R, const float *q_pav, const float *gap, const float *pc, int columnscore, float *p,
                float *tr_out, float *pav);

/* ---- HHblits prefilter kernels, uint8 saturating arithmetic ------------------------------
 * profile: plain [220][Lq] bytes (219 column states + ANY), seq: db column-state sequence (values 0..219).
 * hho_ungapped_score restates Prefilter::ungapped_sse_score (src/hhprefilter.cpp:214-278);
 * hho_sw_score restates Prefilter::swStripedByte (src/hhprefilter.cpp:70-212) literally - striped buffers, the same
 * loops, the same lazy-F exit test - because its result depends on the SIMD striping: E(i,j+1) is updated from the H
 * that contains only the F contributions propagated inside a stripe segment (the lazy-F correction does not touch E),
 * and the lazy-F loop prunes the F chain with a test that equals the full recurrence only for gap_init >= gap_extend.
 * vec_bytes = 32 for the AVX2 build the oracle is pinned to (W = ceil(Lq / vec_bytes)). */
int hho_ungapped_score(const unsigned char *profile, int Lq, const unsigned char *seq, int Ldb, int score_offset);
int hho_sw_score(const unsigned char *profile, int Lq, const unsigned char *seq, int Ldb, int gap_init, int gap_extend,
                 int bias, int vec_bytes);

/* ---- MAC realignment: PosteriorDecoder::realign, src/hhposteriordecoder.cpp:86-119 -----------------
 * Inputs are the prepared p arrays ((L+1)*20) and LINEAR transitions ((L+1)*7, after Log2LinTransitionProbs and the
 * boundary assignments of initializeQueryHMMTransitions / initializeForAlignment); matrices are (Lq+1)*(Lt+1) row-major.
 *   hho_mac_celloff   Viterbi::InitializeForAlignment (non-self, src/hhviterbi.cpp:337-357) + maskViterbiAlignment
 *                     (:205-240) + excludeMACAlignment (:245-262): the cell-off mask realign() builds
 *   hho_mac_forward   forwardAlgorithm (src/hhforwardalgorithm.cpp:10-160): scaled F_MM as float, scale[Lq+2], Pforward
 *   hho_mac_backward  backwardAlgorithm (src/hhbackwardalgorithm.cpp:10-135): turns `fwd` into posteriors in place
 *   hho_mac_dp        macAlgorithm (src/hhmacalgorithm.cpp:18-179): backtrace codes + end point
 *   hho_mac_backtrace backtraceMAC (src/hhbacktracemac.cpp:113-205): path, per-step S and posterior, sum_of_probs
 * Secondary-structure scoring is not restated (ssm2 = 0: ScoreSS = 0 and fpow2(0) = 1 exactly). */
int hho_mac_celloff(int Lq, int Lt, int par_min_overlap, int vi1, int vj1, int vi2, int vj2, int v_nsteps, const int *v_i,
                    const int *v_j, int n_prev, const int *prev_off, const int *prev_i, const int *prev_j, unsigned char *mask);
int hho_mac_forward(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                    float shift, const unsigned char *celloff, float *fwd, double *scale, double *Pforward);
int hho_mac_backward(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                     float shift, const unsigned char *celloff, const double *scale, double Pforward, float *post);
/* The sparse lists of the reference's -o_matrices output as dense planes ((Lq+1)*(Lt+1) floats: the entry's value where the
 * reference pushes one, 0 elsewhere; row-major order = the reference's sorted lists, src/hhbacktracemac.cpp:52-80):
 *   hho_mac_forward_list   src/hhforwardalgorithm.cpp:184-219, on forward's fwd / scale / Pforward (before backward)
 *   hho_mac_backward_list  hho_mac_backward + src/hhbackwardalgorithm.cpp:31-36,112-122 */
int hho_mac_forward_list(const float *fwd, int Lq, int Lt, const double *scale, double Pforward, float *list);
int hho_mac_backward_list(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                          float shift, const unsigned char *celloff, const double *scale, double Pforward, float *post,
                          float *blist);
int hho_mac_dp(const float *post, const unsigned char *celloff, int Lq, int Lt, int local, float mact, unsigned char *bmm,
               int *i2, int *j2);
int hho_mac_backtrace(unsigned char *bmm, const float *post, const float *qp, const float *tp, int Lq, int Lt, int i2, int j2,
                      int *i_steps, int *j_steps, signed char *states, float *S, float *P, int *nsteps, int *matched_cols,
                      float *sum_of_probs);

/* Convenience for the CPU baseline ("port" kind): N templates, score/i2/j2 only, OpenMP over
 * templates.  Returns wall seconds. */
double hho_bench_align(const hho_params *par, const float *qp, const float *qtr, int Lq, int N, const int *L,
                       const float *const *p, const float *const *tr, int threads, float *score, int *i2, int *j2);

/* The same with backtrace + Hit score per template and two path checksums (see oracle/ref_harness.cpp ref_bench_hits). */
double hho_bench_hits(const hho_params *par, const float *qp, const float *qtr, int Lq, int N, const int *L,
                      const float *const *p, const float *const *tr, int threads, float *score, int *i2, int *j2, int *i1,
                      int *j1, int *nsteps, int *matched_cols, float *hit_score, unsigned long long *path_hash,
                      unsigned long long *s_hash);

#ifdef __cplusplus
}
#endif
#endif
