/* oracle/hhv_oracle.c -- TEST INFRASTRUCTURE: plain-C restatement of the reference hot path.
 *
 * See hhv_oracle.h for the rules (checker only; never linked into or called by the product).
 * Compile with -ffp-contract=off: the reference's pinned build has no fused multiply-adds
 * (SURVEY.md 8c), every binary fp32 operation below is rounded on its own, left to right.
 *
 * Restated reference code (file:line in /root/reference):
 *   hho_log2f4               src/hhutil-inl.h:501-541   (LOG_POLY_DEGREE 4)
 *   hho_fast_log2            src/util-inl.h:108-130
 *   hho_dot20_vec            src/hhviterbi.h:126-161    (Viterbi::ScalarProd20Vec)
 *   hho_dot20_scalar         src/hhhit-inl.h:125-131    (ScalarProd20, the branch every build takes)
 *   hho_align                src/hhviterbialgorithm.cpp:29-497 (+ batch padding of
 *                            src/hhhmmsimd.cpp:137-152)
 *   hho_backtrace            src/hhviterbi.cpp:83-160
 *   hho_score_for_backtrace  src/hhviterbi.cpp:195-281, src/hhviterbi.h:193-211 (ScoreSS)
 *   hho_exclude_alignment    src/hhviterbi.cpp:61-77    (VITERBI_PATH_WIDTH = 40, src/hhdecl.h:50)
 */
#include "hhv_oracle.h"

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint32_t f2u(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}
static inline float u2f(uint32_t u) {
  float x;
  memcpy(&x, &u, 4);
  return x;
}
/* x86 MAXPS(a,b): a > b ? a : b (returns b when equal or unordered) */
static inline float mx(float a, float b) { return a > b ? a : b; }

static int g_emission_mode = 0; /* hho_set_emission_mode below */

/* src/hhutil-inl.h:509-541 */
float hho_log2f4(float x) {
  const uint32_t i = f2u(x);
  const float e = (float)((int32_t)((i & 0x7F800000u) >> 23) - 127);
  const float m = u2f((i & 0x007FFFFFu) | 0x3F800000u);
  if (g_emission_mode == 2) { /* the engine's opt-in fused build (viterbi_lane.h HHV_EMISSION_FMA), restated */
    float f = fmaf(-0.107254423828329604454f, m, 0.688243882994381274313f);
    f = fmaf(f, m, -1.75647175389045657003f);
    f = fmaf(f, m, 2.61761038894603480148f);
    return fmaf(f, m - 1.0f, e);
  }
  /* POLY3(m, c0, c1, c2, c3): ((c3*m + c2)*m + c1)*m + c0, each step mul then add */
  float p = -0.107254423828329604454f * m;
  p = p + 0.688243882994381274313f;
  p = p * m;
  p = p + -1.75647175389045657003f;
  p = p * m;
  p = p + 2.61761038894603480148f;
  p = p * (m - 1.0f);
  return p + e;
}

/* src/util-inl.h:108-130.  The lookup table lives in function-local statics that the FIRST caller in the
 * process initialises, and `log(float(1024+i))` is compiled per translation unit: it binds to double
 * log(double) inside hhhmm.cpp but to logf inside hhviterbi.cpp.  In every real hhsearch/hhblits/hhalign run
 * the first call is HMM::AddTransitionPseudocounts (src/hhhmm.cpp:1722-1806, from PrepareQueryHMM,
 * src/hhfunc.cpp:129), long before the Viterbi stage - so the table ScoreForBacktrace sees is the
 * double-log one restated here.  oracle/_ref reproduces that call order (ref_init_fast_log2_like_hhsearch) and
 * tests/test_oracle_vs_reference.py::test_fast_log2_table pins all 1024 entries. */
static float lg2_tab[1025];
static float diff_tab[1025];
static int lg2_init = 0;
static void fast_log2_init(void) {
  float prev = 0.0f;
  lg2_tab[0] = 0.0f;
  for (int i = 1; i <= 1024; ++i) {
    lg2_tab[i] = (float)(log((double)(float)(1024 + i)) * 1.442695041 - (double)10.0f);
    diff_tab[i - 1] = (float)((double)(lg2_tab[i] - prev) * 1.2352E-4);
    prev = lg2_tab[i];
  }
  lg2_init = 1;
}
float hho_fast_log2(float x) {
  if (x <= 0) return -100000;
  if (!lg2_init) {
#pragma omp critical(hho_lg2)
    {
      if (!lg2_init) fast_log2_init();
    }
  }
  const uint32_t u = f2u(x);
  const int a = (int)((u & 0x7F800000u) >> 23) - 0x7f;
  const int b = (int)((u & 0x007FE000u) >> 13);
  const int c = (int)(u & 0x00001FFFu);
  return ((float)a + lg2_tab[b]) + diff_tab[b] * (float)c;
}

/* SWITCH, never touched by a parity test of the default engine.  Mode 0 (default) is the reference's arithmetic.
 * Mode 1 (study, tools/mfma_emission_study.py): the 20-term product as ONE fmaf chain over k - what v_mfma_f32_32x32x2_f32
 * would compute (MI355X_MICROARCH.md, matrix cores).
 * Mode 2: the arithmetic of the engine's OPT-IN fused build (libhhviterbi_hip_fma.so, viterbi_lane.h HHV_EMISSION_FMA): the
 * reference's four partial sums, each accumulating step a fused multiply-add, and log2f4's polynomial fused - so that the
 * opt-in build has an exact checker of its own (tests/test_gpu_fast_mode.py) next to the tolerance comparison with mode 0. */
void hho_set_emission_mode(int mode) { g_emission_mode = mode; }
static float dot20_fma_chain(const float *q, const float *t) {
  float acc = 0.0f;
  for (int k = 0; k < 20; k++) acc = fmaf(q[k], t[k], acc);
  return acc;
}

/* src/hhviterbi.h:126-161 */
float hho_dot20_vec(const float *q, const float *t) {
  if (g_emission_mode == 1) return dot20_fma_chain(q, t);
  float r0 = t[0] * q[0];
  float r1 = t[1] * q[1];
  float r2 = t[2] * q[2];
  float r3 = t[3] * q[3];
  if (g_emission_mode == 2) {
    for (int k = 4; k < 20; k += 4) {
      r0 = fmaf(t[k + 0], q[k + 0], r0);
      r1 = fmaf(t[k + 1], q[k + 1], r1);
      r2 = fmaf(t[k + 2], q[k + 2], r2);
      r3 = fmaf(t[k + 3], q[k + 3], r3);
    }
    r0 = r0 + r1;
    r2 = r2 + r3;
    return r0 + r2;
  }
  for (int k = 4; k < 20; k += 4) {
    r0 = t[k + 0] * q[k + 0] + r0;
    r1 = t[k + 1] * q[k + 1] + r1;
    r2 = t[k + 2] * q[k + 2] + r2;
    r3 = t[k + 3] * q[k + 3] + r3;
  }
  r0 = r0 + r1;
  r2 = r2 + r3;
  return r0 + r2;
}

/* src/hhhit-inl.h:125-131.  The intrinsics branch above it (:85-123) is guarded by `#ifdef SSE`,
 * a macro that no header or build file of the reference defines (lib/simd/simd.h only defines
 * AVX2/AVX512 and the SSE_* size constants), so every build runs this plain left-to-right sum.
 * Pinned against the compiled reference in tests/test_oracle_vs_reference.py. */
float hho_dot20_scalar(const float *q, const float *t) {
  float r = t[0] * q[0];
  for (int k = 1; k < 20; k++) r = r + t[k] * q[k];
  return r;
}

static float ss_cell(const hho_params *par, const hho_ss *ss, int i, int j, int Lt) {
  if (!ss || ss->ss_hmm_mode == HHO_NO_SS) return 0.0f;
  /* index tables of HMMSimd::MapHMMVector, src/hhhmmsimd.cpp:132-135,150-151 (padding -> 0) */
  unsigned char pred_index = 0, dssp_index = 0;
  if (j <= Lt) {
    pred_index = (unsigned char)((unsigned char)ss->t_ss_pred[j] * HHO_MAXCF + ss->t_ss_conf[j]);
    dssp_index = (unsigned char)ss->t_ss_dssp[j];
  }
  const float *score;
  unsigned char idx;
  if (ss->ss_hmm_mode == HHO_PRED_PRED) {
    score = ss->S33 + (((size_t)ss->q_ss_pred[i] * HHO_MAXCF + ss->q_ss_conf[i]) * HHO_NSSPRED * HHO_MAXCF);
    idx = pred_index;
  } else if (ss->ss_hmm_mode == HHO_DSSP_PRED) {
    score = ss->S73 + ((size_t)ss->q_ss_dssp[i] * HHO_NSSPRED * HHO_MAXCF);
    idx = pred_index;
  } else {
    score = ss->S37 + (((size_t)ss->q_ss_pred[i] * HHO_MAXCF + ss->q_ss_conf[i]) * HHO_NDSSP);
    idx = dssp_index;
  }
  return par->ssw * score[idx];
}

int hho_align(const hho_params *par, const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr,
              int Lt, int Lbatch, const unsigned char *celloff, const hho_ss *ss, float *score_out, int *i2_out,
              int *j2_out, unsigned char *bt) {
  if (Lbatch < Lt) return -1;
  const int W = Lbatch + 1;
  const float smin = par->local ? 0.0f : -FLT_MAX; /* :77 */
  const int use_ss = (par->ss_mode == 2 && ss && ss->ss_hmm_mode != HHO_NO_SS); /* src/hhviterbi.cpp:175 */
  float *buf = (float *)malloc(sizeof(float) * 5 * (size_t)W);
  if (!buf) return -2;
  float *sMM = buf, *sDG = buf + W, *sMI = buf + 2 * W, *sGD = buf + 3 * W, *sIM = buf + 4 * W;
  static const float zero20[20] = {0};
  if (bt) memset(bt, 0, (size_t)(Lq + 1) * W);

  /* :144-153 */
  for (int j = 0; j <= Lbatch; ++j) {
    sMM[j] = (float)(-j) * par->egt;
    sDG[j] = sMI[j] = sGD[j] = sIM[j] = -FLT_MAX;
  }
  float score = -FLT_MAX;
  int bi = 0, bj = 0;
  float mm_ij = 0.0f; /* :134 sMM_i_j = simdf32_set(0) */
  int j = 0;
  for (int i = 1; i <= Lq; ++i) {
    /* :161-173 */
    float d_MM = (float)(-(i - 1)) * par->egq;
    float d_IM = -FLT_MAX, d_MI = -FLT_MAX, d_DG = -FLT_MAX, d_GD = -FLT_MAX;
    sMM[0] = (float)(-i) * par->egq;
    sDG[0] = sMI[0] = sGD[0] = sIM[0] = -FLT_MAX;
    /* :182-188 */
    const float q_m2m = qtr[(i - 1) * 7 + HHO_M2M];
    const float q_m2d = qtr[(i - 1) * 7 + HHO_M2D];
    const float q_d2m = qtr[(i - 1) * 7 + HHO_D2M];
    const float q_d2d = qtr[(i - 1) * 7 + HHO_D2D];
    const float q_i2m = qtr[(i - 1) * 7 + HHO_I2M];
    const float q_i2i = qtr[i * 7 + HHO_I2I];
    const float q_m2i = qtr[i * 7 + HHO_M2I];
    const int findMax = (par->local || i == Lq); /* :192 */
    for (j = 1; j <= Lbatch; ++j) {
      /* :222-228 with the padding of src/hhhmmsimd.cpp:137-152 */
      float t_m2m, t_m2d, t_d2m, t_d2d, t_i2m, t_i2i, t_m2i;
      if (j - 1 <= Lt) {
        const float *r = ttr + (size_t)(j - 1) * 7;
        t_m2m = r[HHO_M2M];
        t_m2d = r[HHO_M2D];
        t_d2m = r[HHO_D2M];
        t_d2d = r[HHO_D2D];
        t_i2m = r[HHO_I2M];
      } else {
        t_m2m = t_m2d = t_d2m = t_d2d = t_i2m = -FLT_MAX;
      }
      if (j <= Lt) {
        t_i2i = ttr[(size_t)j * 7 + HHO_I2I];
        t_m2i = ttr[(size_t)j * 7 + HHO_M2I];
      } else {
        t_i2i = t_m2i = -FLT_MAX;
      }
      const float *tpj = (j <= Lt) ? tp + (size_t)j * 20 : zero20;

      /* :241-273 */
      unsigned char b;
      const float c1 = (d_MM + q_m2m) + t_m2m;
      b = (c1 > smin) ? 2 : 0;
      float mm = mx(smin, c1);
      const float c2 = (d_GD + q_m2m) + t_d2m;
      if (c2 > mm) b |= 3;
      mm = mx(mm, c2);
      const float c3 = (d_IM + q_i2m) + t_m2m;
      if (c3 > mm && b < 4) b = 4;
      mm = mx(mm, c3);
      const float c4 = (d_DG + q_d2m) + t_m2m;
      if (c4 > mm && b < 5) b = 5;
      mm = mx(mm, c4);
      const float c5 = (d_MI + q_m2m) + t_i2m;
      if (c5 > mm && b < 6) b = 6;
      mm = mx(mm, c5);

      /* :277-283 */
      float Si = hho_log2f4(hho_dot20_vec(qp + (size_t)i * 20, tpj));
      if (use_ss) Si = ss_cell(par, ss, i, j, Lt) + Si;
      Si = Si + par->shift;
      mm = mm + Si;

      /* :288-298: neighbours from the row buffer, diagonal carries read before the overwrite */
      const float l_MM = sMM[j - 1], l_GD = sGD[j - 1], l_IM = sIM[j - 1];
      const float u_MM = sMM[j], u_DG = sDG[j], u_MI = sMI[j];
      d_MM = sMM[j];
      d_DG = sDG[j];
      d_MI = sMI[j];
      d_GD = sGD[j];
      d_IM = sIM[j];

      /* :307-366 */
      float a, c;
      a = l_MM + t_m2d;
      c = l_GD + t_d2d;
      if (a > c) b ^= 8;
      float gd = mx(a, c);
      a = (l_MM + q_m2i) + t_m2m;
      c = (l_IM + q_i2i) + t_m2m;
      if (a > c) b ^= 16;
      float im = mx(a, c);
      a = u_MM + q_m2d;
      c = u_DG + q_d2d;
      if (a > c) b ^= 32;
      float dg = mx(a, c);
      a = (u_MM + q_m2m) + t_m2i;
      c = (u_MI + q_m2m) + t_i2i;
      if (a > c) b ^= 64;
      float mi = mx(a, c);

      /* :373-392 (only the -DVITERBI_CELLOFF build, i.e. when the matrix carries a mask) */
      if (celloff) {
        const int off = (j <= Lt) ? celloff[(size_t)i * (Lt + 1) + j] : 0;
        const float add = off ? -FLT_MAX : 0.0f;
        mm = mm + add;
        gd = gd + add;
        im = im + add;
        dg = dg + add;
        mi = mi + add;
      }

      /* :396-417 */
      sMM[j] = mm;
      sDG[j] = dg;
      sMI[j] = mi;
      sGD[j] = gd;
      sIM[j] = im;
      if (bt) bt[(size_t)i * W + j] = b;

      /* :423-455 */
      if (findMax) {
        if (mm > score) {
          score = mm;
          bi = i;
          bj = j;
        }
      }
      mm_ij = mm;
    }
    /* :462-486 (j-1 == last column of the batch) */
    if (!par->local) {
      if (mm_ij > score) {
        score = mm_ij;
        bi = i;
        bj = j - 1;
      }
    }
  }
  free(buf);
  *score_out = score;
  *i2_out = bi;
  *j2_out = bj;
  return 0;
}

/* src/hhviterbi.cpp:83-160 */
int hho_backtrace(const unsigned char *bt, int pitch, int i2, int j2, int *i_steps, int *j_steps, signed char *states,
                  int cap, int *nsteps, int *matched_cols) {
  int step = 0, matched = 0;
  int i = i2, j = j2;
  int state = HHO_MM;
#define BT(ii, jj) bt[(size_t)(ii) * pitch + (jj)]
  while (state != HHO_STOP) {
    step++;
    if (step >= cap) return -1;
    states[step] = (signed char)state;
    i_steps[step] = i;
    j_steps[step] = j;
    switch (state) {
      case HHO_MM:
        matched++;
        if (i <= 1 || j <= 1) {
          state = HHO_STOP;
        } else {
          state = BT(i, j) & 7;
          i--;
          j--;
        }
        break;
      case HHO_GD:
        if (j <= 1) state = HHO_STOP;
        else {
          if (BT(i, j) & 8) state = HHO_MM;
          j--;
        }
        break;
      case HHO_IM:
        if (j <= 1) state = HHO_STOP;
        else {
          if (BT(i, j) & 16) state = HHO_MM;
          j--;
        }
        break;
      case HHO_DG:
        if (i <= 1) state = HHO_STOP;
        else {
          if (BT(i, j) & 32) state = HHO_MM;
          i--;
        }
        break;
      case HHO_MI:
        if (i <= 1) state = HHO_STOP;
        else {
          if (BT(i, j) & 64) state = HHO_MM;
          i--;
        }
        break;
      default: /* :139-144 illegal state value: stop */
        state = HHO_STOP;
        break;
    }
  }
#undef BT
  states[step] = HHO_MM; /* :147 */
  *nsteps = step;
  *matched_cols = matched;
  return 0;
}

static float score_ss_cell(const hho_params *par, const hho_ss *ss, int i, int j) {
  /* src/hhviterbi.h:193-211 (ScoreSS with ssm = ss_hmm_mode) */
  if (!ss) return 0.0f;
  switch (ss->ss_hmm_mode) {
    case HHO_PRED_DSSP:
      return par->ssw * ss->S37[((size_t)ss->q_ss_pred[i] * HHO_MAXCF + ss->q_ss_conf[i]) * HHO_NDSSP + ss->t_ss_dssp[j]];
    case HHO_DSSP_PRED:
      return par->ssw *
             ss->S73[((size_t)ss->q_ss_dssp[i] * HHO_NSSPRED + ss->t_ss_pred[j]) * HHO_MAXCF + ss->t_ss_conf[j]];
    case HHO_PRED_PRED:
      return par->ssw * ss->S33[(((size_t)ss->q_ss_pred[i] * HHO_MAXCF + ss->q_ss_conf[i]) * HHO_NSSPRED +
                                 ss->t_ss_pred[j]) * HHO_MAXCF + ss->t_ss_conf[j]];
    default:
      return 0.0f;
  }
}

/* src/hhviterbi.cpp:195-281 */
int hho_score_for_backtrace(const hho_params *par, const float *qp, const float *tp, const hho_ss *ss,
                            const int *i_steps, const int *j_steps, const signed char *states, int nsteps,
                            float viterbi_score, float *S, float *hit_score, float *score_ss_out) {
  float score_ss = 0.0f;
  float score = viterbi_score;
  for (int step = 1; step <= nsteps; step++) {
    if (states[step] == HHO_MM) {
      S[step] = hho_fast_log2(hho_dot20_scalar(qp + (size_t)i_steps[step] * 20, tp + (size_t)j_steps[step] * 20));
      score_ss += score_ss_cell(par, ss, i_steps[step], j_steps[step]);
    } else {
      S[step] = 0.0f;
    }
  }
  if (par->ss_mode == 2) score -= score_ss;
  float Scorr = 0;
  if (nsteps) {
    for (int step = 2; step <= nsteps; step++) Scorr += S[step] * S[step - 1];
    for (int step = 3; step <= nsteps; step++) Scorr += S[step] * S[step - 2];
    for (int step = 4; step <= nsteps; step++) Scorr += S[step] * S[step - 3];
    for (int step = 5; step <= nsteps; step++) Scorr += S[step] * S[step - 4];
    score += par->corr * Scorr;
  }
  *hit_score = score;
  *score_ss_out = score_ss;
  return 0;
}

/* src/hhviterbi.cpp:61-77 */
int hho_exclude_alignment(int Lq, int Lt, const int *i_steps, const int *j_steps, int nsteps, unsigned char *mask) {
  const int PW = 40;
  for (int step = 1; step < nsteps; step++) {
    const int i = i_steps[step], j = j_steps[step];
    const int ilo = i - PW > 1 ? i - PW : 1, ihi = i + PW < Lq ? i + PW : Lq;
    for (int ii = ilo; ii <= ihi; ++ii) mask[(size_t)ii * (Lt + 1) + j] = 1;
    const int jlo = j - PW > 1 ? j - PW : 1, jhi = j + PW < Lt ? j + PW : Lt;
    for (int jj = jlo; jj <= jhi; ++jj) mask[(size_t)i * (Lt + 1) + jj] = 1;
  }
  return 0;
}

/* src/util-inl.h:190-215 */
float hho_fpow2(float x) {
  if (x >= FLT_MAX_EXP) return FLT_MAX;
  if (x <= FLT_MIN_EXP) return 0.0f;
  const float tx = (x - 0.5f) + (float)(3 << 22);
  const int lx = (int)(f2u(tx) - 0x4b400000u);
  const float dx = x - (float)lx;
  float y = dx * 0.0134929f;
  y = 0.0520749f + y;
  y = dx * y;
  y = 0.241404f + y;
  y = dx * y;
  y = 0.693019f + y;
  y = dx * y;
  y = 1.0f + y;
  return u2f(f2u(y) + ((uint32_t)lx << 23));
}

int hho_prepare(int role, int L, const float *f, const float *tr, const float *neff, float Neff_HMM, const float *pb,
                const float *R, const float *q_pav, const float *gap, const float *pc, int columnscore, float *p,
                float *tr_out, float *pav) {
  const float gapd = gap[0], gape = gap[1], gapf = gap[2], gapg = gap[3], gaph = gap[4], gapi = gap[5], gapb = gap[6];
  memcpy(tr_out, tr, sizeof(float) * 7 * (size_t)(L + 1));
  /* ---- AddTransitionPseudocounts, src/hhhmm.cpp:1722-1806 */
  if (gapb > 0) {
    float pM2D, pM2I;
    pM2D = pM2I = (float)((double)gapd * 0.0286);
    const float pM2M = 1 - pM2D - pM2I;
    const float pI2I = (float)(1.0 * (double)gape / ((double)(gape - 1) + 1.0 / 0.75));
    const float pI2M = 1 - pI2I;
    const float pD2D = (float)(1.0 * (double)gape / ((double)(gape - 1) + 1.0 / 0.75));
    const float pD2M = 1 - pD2D;
    for (int i = 0; i <= L; ++i) {
      float *t = tr_out + (size_t)i * 7;
      const float nM = neff[i * 3 + 0], nI = neff[i * 3 + 1], nD = neff[i * 3 + 2];
      float p0 = (nM - 1) * hho_fpow2(t[HHO_M2M]) + gapb * pM2M;
      float p1 = (nM - 1) * hho_fpow2(t[HHO_M2D]) + gapb * pM2D;
      float p2 = (nM - 1) * hho_fpow2(t[HHO_M2I]) + gapb * pM2I;
      if (i == 0) p1 = p2 = 0;
      if (i == L) p1 = p2 = 0;
      float sum = p0 + p1 + p2 + FLT_MIN;
      t[HHO_M2M] = hho_fast_log2(p0 / sum);
      t[HHO_M2D] = hho_fast_log2(p1 / sum) * gapf;
      t[HHO_M2I] = hho_fast_log2(p2 / sum) * gapg;
      p0 = nI * hho_fpow2(t[HHO_I2M]) + gapb * pI2M;
      p1 = nI * hho_fpow2(t[HHO_I2I]) + gapb * pI2I;
      sum = p0 + p1 + FLT_MIN;
      t[HHO_I2M] = hho_fast_log2(p0 / sum);
      t[HHO_I2I] = hho_fast_log2(p1 / sum) * gapi;
      p0 = nD * hho_fpow2(t[HHO_D2M]) + gapb * pD2M;
      p1 = nD * hho_fpow2(t[HHO_D2D]) + gapb * pD2D;
      if (i == L) p1 = 0;
      sum = p0 + p1 + FLT_MIN;
      t[HHO_D2M] = hho_fast_log2(p0 / sum);
      t[HHO_D2D] = hho_fast_log2(p1 / sum) * gaph;
    }
  }
  /* ---- PreparePseudocounts (:1811-1815) + AddAminoAcidPseudocounts (:1874-1964) */
  const int pcm = (int)pc[0];
  const float pca = pc[1], pcb = pc[2], pcc = pc[3];
  if (pcm < 0 || pcm > 3) return -1;
  memset(p, 0, sizeof(float) * 20 * (size_t)(L + 2));
  for (int i = 1; i <= L; ++i) {
    const float *fi = f + (size_t)i * 20;
    float tau = 0.0f;
    if (pcm == 1) tau = pca;
    if (pcm == 2) {
      if (pcc == 1.0f) tau = (float)fmin(1.0, (double)pca / (1. + (double)(neff[i * 3] / pcb)));
      /* :1905: pow(Neff_M[i] / pcb, pcc) with float arguments in C++ is the float overload (= powf); its result joins a
         double expression */
      else tau = (float)fmin(1.0, (double)pca / (1. + (double)powf(neff[i * 3] / pcb, pcc)));
    }
    if (pcm == 3) {
      /* :1911-1919 constant-diversity pseudocounts: x and the product are float expressions, pca is recomputed from pcb (a
         double expression stored in the float parameter), fmax(0.0, ..) is the double overload */
      const float x = neff[i * 3] / pcb;
      const float pca3 = (float)(0.793 + 0.048 * ((double)pcb - 10.0));
      tau = (float)fmax(0.0, (double)(pca3 * ((1.0f - x) + (pcc * x) * (1.0f - x))));
    }
    for (int a = 0; a < 20; ++a) {
      if (pcm == 0) {
        p[(size_t)i * 20 + a] = fi[a];
      } else {
        const float g = hho_dot20_scalar(R + a * 20, fi); /* ScalarProd20(R[a], f[i]) */
        p[(size_t)i * 20 + a] = (float)((1. - (double)tau) * (double)fi[a] + (double)(tau * g));
      }
    }
  }
  /* ---- CalculateAminoAcidBackground, :1854-1868 */
  for (int a = 0; a < 20; ++a) pav[a] = pb[a] * 100.0f / Neff_HMM;
  for (int i = 1; i <= L; ++i)
    for (int a = 0; a < 20; ++a) pav[a] += p[(size_t)i * 20 + a];
  {
    float sum = 0.0f;
    for (int k = 0; k < 20; k++) sum += pav[k];
    if (sum != 0.0f) {
      const float fac = (float)(1.0 / (double)sum);
      for (int k = 0; k < 20; k++) pav[k] *= fac;
    }
  }
  for (int a = 0; a < 20; ++a) p[a] = p[(size_t)(L + 1) * 20 + a] = pav[a];
  /* ---- IncludeNullModelInHMM, :2059-2144 */
  if (role == 1) {
    float pnul[20];
    for (int a = 0; a < 20; ++a) {
      switch (columnscore) {
        case 1: pnul[a] = (float)(0.5 * (double)(q_pav[a] + pav[a])); break;
        case 2: pnul[a] = pav[a]; break;
        case 3: pnul[a] = q_pav[a]; break;
        default: pnul[a] = pb[a]; break;
      }
    }
    for (int j = 0; j <= L + 1; ++j)
      for (int a = 0; a < 20; ++a) p[(size_t)j * 20 + a] /= pnul[a];
  }
  return 0;
}

/* ---- prefilter kernels -------------------------------------------------------------------------------------- */
static inline int sat_add8(int a, int b) { return a + b > 255 ? 255 : a + b; }
static inline int sat_sub8(int a, int b) { return a - b < 0 ? 0 : a - b; }
static inline int imax2(int a, int b) { return a > b ? a : b; }

/* src/hhprefilter.cpp:214-278: along every diagonal S = max(0, min(255, S + q(i,x_j)) - offset); the padding
 * positions of the striped profile only carry values forward and never create a new maximum */
int hho_ungapped_score(const unsigned char *profile, int Lq, const unsigned char *seq, int Ldb, int score_offset) {
  int smax = 0;
  unsigned char *prev = (unsigned char *)calloc((size_t)Lq + 1, 1), *cur = (unsigned char *)calloc((size_t)Lq + 1, 1);
  for (int j = 0; j < Ldb; ++j) {
    const unsigned char *q = profile + (size_t)seq[j] * Lq;
    for (int i = 0; i < Lq; ++i) {
      const int diag = i ? prev[i - 1] : 0;
      const int s = sat_sub8(sat_add8(diag, q[i]), score_offset);
      cur[i] = (unsigned char)s;
      smax = imax2(smax, s);
    }
    unsigned char *t = prev;
    prev = cur;
    cur = t;
  }
  free(prev);
  free(cur);
  return smax;
}

/* src/hhprefilter.cpp:70-212, restated position by position (p = query position):
 *   base   = max(0, min(255, H(p-1, j-1) + q(p, x_j)) - bias)
 *   Hpre   = max(base, E(p, j), Fin(p))        Fin: F chain restarted at every stripe segment start (p % W == 0)
 *   E(p, j+1) = max(E(p,j) - ge, Hpre - go)    (NOT from the lazy-F corrected H, :171 comment of the reference)
 *   H(p, j)   = max(Hpre, Ffull(p))            Ffull: the F chain over the whole column (the lazy-F loop)
 * all with unsigned saturation; the result is the maximum H. */
int hho_sw_score(const unsigned char *profile, int Lq, const unsigned char *seq, int Ldb, int gap_init, int gap_extend,
                 int bias, int vec_bytes) {
  /* Literal restatement of the striped loops (src/hhprefilter.cpp:70-212): V = vec_bytes vector elements, element k of
   * segment row j is query position k*W + j.  The lazy-F loop is kept as it is written - its exit test
   * F - (H - gap_init) == 0 prunes the F chain, which equals the full recurrence only for gap_init >= gap_extend. */
  const int V = vec_bytes, W = (Lq + V - 1) / V;
  unsigned char *Hs = (unsigned char *)calloc((size_t)W * V, 1), *Hl = (unsigned char *)calloc((size_t)W * V, 1);
  unsigned char *E = (unsigned char *)calloc((size_t)W * V, 1), *F = (unsigned char *)calloc((size_t)V, 1);
  unsigned char *H = (unsigned char *)calloc((size_t)V, 1);
  int best = 0;
#define QP(x, j, k) (((k) * W + (j)) < Lq ? profile[(size_t)(x) * Lq + (k) * W + (j)] : (unsigned char)bias)
  for (int i = 0; i < Ldb; ++i) {
    const int x = seq[i];
    memset(F, 0, (size_t)V);
    /* vH = pvHStore[W-1] shifted by one element (:120-126) */
    for (int k = V - 1; k >= 1; --k) H[k] = Hs[(size_t)(W - 1) * V + k - 1];
    H[0] = 0;
    unsigned char *t = Hl; /* swap the two H buffers (:129-132) */
    Hl = Hs;
    Hs = t;
    for (int j = 0; j < W; ++j) {
      for (int k = 0; k < V; ++k) {
        int h = sat_sub8(sat_add8(H[k], QP(x, j, k)), bias);
        unsigned char *e = &E[(size_t)j * V + k];
        h = imax2(imax2(h, *e), F[k]);
        best = imax2(best, h);
        Hs[(size_t)j * V + k] = (unsigned char)h;
        const int hg = sat_sub8(h, gap_init);
        *e = (unsigned char)imax2(sat_sub8(*e, gap_extend), hg);
        F[k] = (unsigned char)imax2(sat_sub8(F[k], gap_extend), hg);
        H[k] = Hl[(size_t)j * V + k];
      }
    }
    /* lazy-F loop (:176-203) */
    int j = 0;
    for (int k = 0; k < V; ++k) H[k] = Hs[k];
    for (int k = V - 1; k >= 1; --k) F[k] = F[k - 1];
    F[0] = 0;
    for (;;) {
      int any = 0;
      for (int k = 0; k < V; ++k) any |= sat_sub8(F[k], sat_sub8(H[k], gap_init)) != 0;
      if (!any) break;
      for (int k = 0; k < V; ++k) {
        const int h = imax2(H[k], F[k]);
        best = imax2(best, h);
        Hs[(size_t)j * V + k] = (unsigned char)h;
        F[k] = (unsigned char)sat_sub8(F[k], gap_extend);
      }
      if (++j >= W) {
        j = 0;
        for (int k = V - 1; k >= 1; --k) F[k] = F[k - 1];
        F[0] = 0;
      }
      for (int k = 0; k < V; ++k) H[k] = Hs[(size_t)j * V + k];
    }
  }
#undef QP
  free(Hs);
  free(Hl);
  free(E);
  free(F);
  free(H);
  return best;
}

/* ==== MAC realignment ======================================================================== */
enum { T_M2M = 0, T_M2I = 1, T_M2D = 2, T_I2M = 3, T_I2I = 4, T_D2M = 5, T_D2D = 6 }; /* src/hhdecl.h:68 */
#define MAC_PATHWIDTH 40 /* FWD_BKW_PATHWITDH, src/hhdecl.h:37 */

int hho_mac_celloff(int Lq, int Lt, int par_min_overlap, int vi1, int vj1, int vi2, int vj2, int v_nsteps, const int *v_i,
                    const int *v_j, int n_prev, const int *prev_off, const int *prev_i, const int *prev_j, unsigned char *mask) {
  const int pitch = Lt + 1;
#define CO(i, j) mask[(size_t)(i) * pitch + (j)]
  memset(mask, 0, (size_t)(Lq + 1) * pitch);
  /* Viterbi::InitializeForAlignment, two different HMMs (src/hhviterbi.cpp:337-357) */
  const int lmin = Lq < Lt ? Lq : Lt;
  int mo;
  if (par_min_overlap == 0) {
    mo = (int)(0.333f * lmin) + 1;
    if (mo > 60) mo = 60;
  } else {
    mo = (int)(0.8f * lmin);
    if (par_min_overlap < mo) mo = par_min_overlap;
  }
  for (int i = 0; i < mo; ++i)
    for (int j = i - mo + Lt + 1; j <= Lt; ++j)
      if (j >= 0) CO(i, j) = 1;
  for (int i = Lq - mo + 1; i <= Lq; ++i)
    for (int j = 1; j < i + mo - Lq; ++j)
      if (i >= 0 && j <= Lt) CO(i, j) = 1;
  /* maskViterbiAlignment (src/hhposteriordecoder.cpp:205-240): overwrites every cell 1..Lq x 1..Lt */
  for (int i = 1; i <= Lq; ++i)
    for (int j = 1; j <= Lt; ++j) CO(i, j) = !((i < vi1 && j < vj1) || (i > vi2 && j > vj2));
  for (int step = v_nsteps; step >= 1; --step) {
    const int lo = v_i[step] - MAC_PATHWIDTH > 1 ? v_i[step] - MAC_PATHWIDTH : 1;
    const int hi = v_i[step] + MAC_PATHWIDTH < Lq ? v_i[step] + MAC_PATHWIDTH : Lq;
    for (int i = lo; i <= hi; ++i) CO(i, v_j[step]) = 0;
  }
  for (int step = v_nsteps; step >= 1; --step) {
    const int lo = v_j[step] - MAC_PATHWIDTH > 1 ? v_j[step] - MAC_PATHWIDTH : 1;
    const int hi = v_j[step] + MAC_PATHWIDTH < Lt ? v_j[step] + MAC_PATHWIDTH : Lt;
    for (int j = lo; j <= hi; ++j) CO(v_i[step], j) = 0;
  }
  /* excludeMACAlignment (:245-262) for every earlier MAC alignment of this template */
  for (int k = 0; k < n_prev; ++k)
    for (int s = prev_off[k]; s < prev_off[k + 1]; ++s) {
      const int i = prev_i[s], j = prev_j[s];
      for (int ii = (i - 2 > 1 ? i - 2 : 1); ii <= (i + 2 < Lq ? i + 2 : Lq); ++ii) CO(ii, j) = 1;
      for (int jj = (j - 2 > 1 ? j - 2 : 1); jj <= (j + 2 < Lt ? j + 2 : Lt); ++jj) CO(i, jj) = 1;
    }
  for (int j = 0; j <= Lt; ++j) CO(0, j) = 0; /* row/column 0 are never read by the DP */
  for (int i = 0; i <= Lq; ++i) CO(i, 0) = 0;
#undef CO
  return 0;
}

typedef struct {
  double mm, gd, im, dg, mi;
} mac_col; /* PosteriorMatrixCol, src/hhposteriordecoder.h:75-81 */

int hho_mac_forward(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                    float shift, const unsigned char *celloff, float *fwd, double *scale, double *Pforward) {
  const int pitch = Lt + 1;
  mac_col *curr = (mac_col *)calloc((size_t)Lt + 3, sizeof(mac_col)), *prev = (mac_col *)calloc((size_t)Lt + 3, sizeof(mac_col));
  if (!curr || !prev) return -1;
#define QT(i, a) qtr[(size_t)(i) * 7 + (a)]
#define TT(j, a) ttr[(size_t)(j) * 7 + (a)]
#define PF(i, j) hho_dot20_scalar(qp + (size_t)(i) * 20, tp + (size_t)(j) * 20) /* ProbFwd, src/hhhit-inl.h:125 */
  double pmin = local ? 1.0 : 0.0;
  const double Cshift = pow(2.0, shift);
  double scale_prod = 1.0;
  memset(fwd, 0, sizeof(float) * (size_t)(Lq + 1) * pitch);
  /* row 1 (:21-41) */
  curr[0].mm = curr[0].im = curr[0].gd = 0.0;
  for (int j = 1; j <= Lt; ++j) {
    if (celloff[pitch + j]) {
      curr[j].mm = curr[j].mi = curr[j].dg = curr[j].im = curr[j].gd = 0.0;
    } else {
      curr[j].mm = PF(1, j) * Cshift;
      curr[j].mi = curr[j].dg = 0.0;
      curr[j].im = curr[j - 1].mm * QT(1, T_M2I) * TT(j - 1, T_M2M) + curr[j - 1].im * QT(1, T_I2I) * TT(j - 1, T_M2M);
      curr[j].gd = curr[j - 1].mm * TT(j - 1, T_M2D) + curr[j - 1].gd * TT(j - 1, T_D2D);
    }
  }
  for (int j = 0; j <= Lt; ++j) {
    fwd[pitch + j] = (float)curr[j].mm; /* row 0 of p_mm is never read */
    prev[j] = curr[j];
  }
  scale[0] = scale[1] = scale[2] = 1.0;
  for (int i = 2; i <= Lq; ++i) {
    if (scale_prod < DBL_MIN * 100)
      scale_prod = 0.0;
    else
      scale_prod *= scale[i];
    /* first column (:67-83) */
    if (celloff[(size_t)i * pitch + 1]) {
      curr[1].mm = curr[1].mi = curr[1].dg = curr[1].im = curr[1].gd = 0.0;
    } else {
      curr[1].mm = scale_prod * 1.0f * PF(i, 1) * Cshift; /* fpow2(ScoreSS) = fpow2(0) = 1.0f */
      curr[1].im = curr[1].gd = 0.0;
      curr[1].mi = scale[i] * (prev[1].mm * QT(i - 1, T_M2M) * TT(1, T_M2I) + prev[1].mi * QT(i - 1, T_M2M) * TT(1, T_I2I));
      curr[1].dg = scale[i] * (prev[1].mm * QT(i - 1, T_M2D) + prev[1].dg * QT(i - 1, T_D2D));
    }
    double Pmax = 0;
    memset(curr + 2, 0, (size_t)Lt * sizeof(mac_col));
    for (int j = 2; j <= Lt; ++j) {
      if (celloff[(size_t)i * pitch + j]) continue;
      curr[j].mm = PF(i, j) * Cshift * 1.0f * scale[i] *
                   (pmin + prev[j - 1].mm * QT(i - 1, T_M2M) * TT(j - 1, T_M2M) + prev[j - 1].gd * QT(i - 1, T_M2M) * TT(j - 1, T_D2M) +
                    prev[j - 1].im * QT(i - 1, T_I2M) * TT(j - 1, T_M2M) + prev[j - 1].dg * QT(i - 1, T_D2M) * TT(j - 1, T_M2M) +
                    prev[j - 1].mi * QT(i - 1, T_M2M) * TT(j - 1, T_I2M));
      curr[j].gd = (curr[j - 1].mm * TT(j - 1, T_M2D) + curr[j - 1].gd * TT(j - 1, T_D2D));
      curr[j].im = (curr[j - 1].mm * QT(i, T_M2I) * TT(j - 1, T_M2M) + curr[j - 1].im * QT(i, T_I2I) * TT(j - 1, T_M2M));
      curr[j].dg = scale[i] * (prev[j].mm * QT(i - 1, T_M2D) + prev[j].dg * QT(i - 1, T_D2D));
      curr[j].mi = scale[i] * (prev[j].mm * QT(i - 1, T_M2M) * TT(j, T_M2I) + prev[j].mi * QT(i - 1, T_M2M) * TT(j, T_I2I));
      Pmax = fmax(Pmax, curr[j].mm);
    }
    for (int j = 0; j <= Lt; ++j) fwd[(size_t)i * pitch + j] = (float)curr[j].mm;
    mac_col *tmp = prev;
    prev = curr;
    curr = tmp;
    pmin *= scale[i];
    if (pmin < DBL_MIN * 100) pmin = 0.0;
    scale[i + 1] = 1.0 / (Pmax + 1.0);
  }
  /* total forward probability (:162-182) */
  double Pf;
  if (local) {
    Pf = 1.0;
    for (int i = 1; i <= Lq; ++i) {
      for (int j = 1; j <= Lt; ++j) Pf += fwd[(size_t)i * pitch + j];
      Pf *= scale[i + 1];
    }
  } else {
    Pf = 0.0;
    for (int i = 1; i < Lq; ++i) Pf = (Pf + fwd[(size_t)i * pitch + Lt] * scale[i + 1]);
    for (int j = 1; j <= Lt; ++j) Pf += fwd[(size_t)Lq * pitch + j];
    Pf *= scale[Lq + 1];
  }
  *Pforward = Pf;
  free(curr);
  free(prev);
  return 0;
}

/* PosteriorDecoder::m_back_forward_matrix_threshold (src/hhposteriordecoder.cpp:62): a float */
static const float mac_list_threshold = 0.0001f;

/* "save forward profile" (src/hhforwardalgorithm.cpp:184-219), on forward's outputs and before backward overwrites fwd: the
 * dense plane `list` ((Lq+1)*(Lt+1)) takes (float)ffprob where the reference pushes an entry (ffprob > 1e-4), 0 elsewhere.
 * Entries read row by row, column by column are the reference's sorted list (src/hhbacktracemac.cpp:68-80). */
int hho_mac_forward_list(const float *fwd, int Lq, int Lt, const double *scale, double Pforward, float *list) {
  const int pitch = Lt + 1;
  double scale_prod = 1.0; /* as forward's row loop leaves it (:17, :66-69) */
  for (int i = 2; i <= Lq; ++i) {
    if (scale_prod < DBL_MIN * 100)
      scale_prod = 0.0;
    else
      scale_prod *= scale[i];
  }
  memset(list, 0, sizeof(float) * (size_t)(Lq + 1) * pitch);
  double scale_rate, scale_prod_curr = 1.0, ffprob;
  for (int i = 1; i <= Lq; ++i) {
    if (scale_prod_curr < DBL_MIN * 100)
      scale_prod_curr = 0.0;
    else
      scale_prod_curr *= scale[i];
    for (int j = 1; j <= Lt; ++j) {
      if (scale_prod_curr == 0.0)
        scale_rate = 0.0;
      else
        scale_rate = (scale_prod * scale[Lq + 1]) / scale_prod_curr;
      ffprob = (fwd[(size_t)i * pitch + j] / Pforward) * scale_rate;
      if (ffprob > mac_list_threshold) list[(size_t)i * pitch + j] = (float)ffprob;
    }
  }
  return 0;
}

static int mac_backward_impl(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                             float shift, const unsigned char *celloff, const double *scale, double Pforward, float *post,
                             float *blist);
int hho_mac_backward(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                     float shift, const unsigned char *celloff, const double *scale, double Pforward, float *post) {
  return mac_backward_impl(qp, qtr, Lq, tp, ttr, Lt, local, shift, celloff, scale, Pforward, post, NULL);
}
/* the same, and the backward list (src/hhbackwardalgorithm.cpp:112-122) as a dense plane like hho_mac_forward_list's */
int hho_mac_backward_list(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                          float shift, const unsigned char *celloff, const double *scale, double Pforward, float *post,
                          float *blist) {
  memset(blist, 0, sizeof(float) * (size_t)(Lq + 1) * (Lt + 1));
  return mac_backward_impl(qp, qtr, Lq, tp, ttr, Lt, local, shift, celloff, scale, Pforward, post, blist);
}
static int mac_backward_impl(const float *qp, const float *qtr, int Lq, const float *tp, const float *ttr, int Lt, int local,
                             float shift, const unsigned char *celloff, const double *scale, double Pforward, float *post,
                             float *blist) {
  const int pitch = Lt + 1;
  mac_col *curr = (mac_col *)calloc((size_t)Lt + 3, sizeof(mac_col)), *prev = (mac_col *)calloc((size_t)Lt + 3, sizeof(mac_col));
  if (!curr || !prev) return -1;
  const double Cshift = pow(2.0, shift);
  double scale_prod = scale[Lq + 1];
  for (int j = Lt; j >= 1; --j) {
    float *pv = post + (size_t)Lq * pitch + j;
    if (celloff[(size_t)Lq * pitch + j]) {
      *pv = 0.0f;
      prev[j].mm = 0.0;
    } else {
      prev[j].mm = scale[Lq + 1];
      *pv = (float)(*pv * scale[Lq + 1] / Pforward);
    }
    prev[j].mi = prev[j].dg = 0.0;
  }
  double final_scale_prod = scale[Lq + 1]; /* :31-36 */
  for (int i = Lq - 1; i >= 1; i--) {
    final_scale_prod *= scale[i + 1];
    if (final_scale_prod < DBL_MIN * 100) final_scale_prod = 0.0;
  }
  double pmin = local ? scale[Lq + 1] : 0.0;
  for (int i = Lq - 1; i >= 1; --i) {
    scale_prod *= scale[i + 1];
    if (scale_prod < DBL_MIN * 100) scale_prod = 0.0;
    float *row = post + (size_t)i * pitch;
    if (celloff[(size_t)i * pitch + Lt]) {
      row[Lt] = 0.0f;
      curr[Lt].mm = 0.0;
    } else {
      curr[Lt].mm = scale_prod;
      row[Lt] = (float)(row[Lt] * scale_prod / Pforward);
    }
    pmin *= scale[i + 1];
    if (pmin < DBL_MIN * 100) pmin = 0.0;
    curr[Lt].im = curr[Lt].mi = curr[Lt].dg = curr[Lt].gd = 0.0;
    if (Lt > 1) memset(curr + 1, 0, (size_t)(Lt - 1) * sizeof(mac_col));
    for (int j = Lt - 1; j >= 1; --j) {
      if (celloff[(size_t)i * pitch + j]) continue;
      const double pmatch = prev[j + 1].mm * PF(i + 1, j + 1) * 1.0f * Cshift * scale[i + 1];
      curr[j].mm = (+pmin + pmatch * QT(i, T_M2M) * TT(j, T_M2M) + curr[j + 1].gd * TT(j, T_M2D) +
                    curr[j + 1].im * QT(i, T_M2I) * TT(j, T_M2M) + prev[j].dg * QT(i, T_M2D) * scale[i + 1] +
                    prev[j].mi * QT(i, T_M2M) * TT(j, T_M2I) * scale[i + 1]);
      curr[j].gd = (+pmatch * QT(i, T_M2M) * TT(j, T_D2M) + curr[j + 1].gd * TT(j, T_D2D));
      curr[j].im = (+pmatch * QT(i, T_I2M) * TT(j, T_M2M) + curr[j + 1].im * QT(i, T_I2I) * TT(j, T_M2M));
      curr[j].dg = (+pmatch * QT(i, T_D2M) * TT(j, T_M2M) + prev[j].dg * QT(i, T_D2D) * scale[i + 1]);
      curr[j].mi = (+pmatch * QT(i, T_M2M) * TT(j, T_I2M) + prev[j].mi * QT(i, T_M2M) * TT(j, T_I2I) * scale[i + 1]);
      if (blist) { /* :112-122 */
        float substitutionScore = PF(i, j);
        float actual_backward_single = substitutionScore * Cshift * curr[j].mm / Pforward * final_scale_prod / scale_prod;
        if (actual_backward_single > mac_list_threshold) blist[(size_t)i * pitch + j] = actual_backward_single;
      }
    }
    /* multiplyPosteriorValue takes a float: F (float) * (float)(B / Pforward) (:122-124, hhposteriormatrix.h:43) */
    for (int j = 1; j <= Lt - 1; ++j) row[j] *= (float)(curr[j].mm / Pforward);
    mac_col *tmp = prev;
    prev = curr;
    curr = tmp;
  }
  free(curr);
  free(prev);
  return 0;
}
#undef QT
#undef TT
#undef PF

enum { MAC_STOP = 0, MAC_MM = 2, MAC_IM = 4, MAC_MI = 6 }; /* ViterbiMatrix::STOP/MM/IM/MI, src/hhviterbimatrix.h */

int hho_mac_dp(const float *post, const unsigned char *celloff, int Lq, int Lt, int local, float mact, unsigned char *bmm,
               int *i2, int *j2) {
  const int pitch = Lt + 1;
  float *S_prev = (float *)calloc((size_t)Lt + 2, sizeof(float)), *S_curr = (float *)calloc((size_t)Lt + 2, sizeof(float));
  if (!S_prev || !S_curr) return -1;
  float score_MAC = -FLT_MAX;
  *i2 = *j2 = 0;
  memset(bmm, 0, (size_t)(Lq + 1) * pitch);
  bmm[0] = MAC_STOP;
  for (int i = 1; i <= Lq; ++i) {
    const int jmax = Lt; /* hit.min_overlap = 0 (:54): jmin = 1, jmax = t.L */
    S_curr[0] = 0.0;
    for (int j = 1; j <= Lt; ++j) {
      if (celloff[(size_t)i * pitch + j]) {
        S_curr[j] = -FLT_MIN;
        bmm[(size_t)i * pitch + j] = MAC_STOP;
        continue;
      }
      const float p = post[(size_t)i * pitch + j];
      const float term1 = p - mact;
      const float term2 = S_prev[j - 1] + p - mact;
      const float term3 = (float)(S_prev[j] - 0.5 * mact);
      const float term4 = (float)(S_curr[j - 1] - 0.5 * mact);
      float mxv;
      unsigned char val;
      if (term1 > term2) {
        mxv = term1;
        val = MAC_STOP;
      } else {
        mxv = term2;
        val = MAC_MM;
      }
      if (term3 > mxv) {
        mxv = term3;
        val = MAC_MI;
      }
      if (term4 > mxv) {
        mxv = term4;
        val = MAC_IM;
      }
      S_curr[j] = mxv;
      bmm[(size_t)i * pitch + j] = val;
      if (mxv > score_MAC && (local || i == Lq)) {
        *i2 = i;
        *j2 = j;
        score_MAC = mxv;
      }
    }
    if (!local && S_curr[jmax] > score_MAC) {
      *i2 = i;
      *j2 = jmax;
      score_MAC = S_curr[jmax];
    }
    for (int j = 0; j <= Lt; ++j) S_prev[j] = S_curr[j];
  }
  free(S_prev);
  free(S_curr);
  return 0;
}

int hho_mac_backtrace(unsigned char *bmm, const float *post, const float *qp, const float *tp, int Lq, int Lt, int i2, int j2,
                      int *i_steps, int *j_steps, signed char *states, float *S, float *P, int *nsteps, int *matched_cols,
                      float *sum_of_probs) {
  const int pitch = Lt + 1;
  for (int i = 0; i <= Lq; ++i) bmm[(size_t)i * pitch + 1] = MAC_STOP; /* :124-125 */
  for (int j = 1; j <= Lt; ++j) bmm[pitch + j] = MAC_STOP;
  int matched = 1, step = 0, i = i2, j = j2, state = MAC_MM;
  if (bmm[(size_t)i * pitch + j] != MAC_MM) {
    step = 0;
    i_steps[0] = i;
    j_steps[0] = j;
  } else {
    while (state != MAC_STOP) {
      step++;
      states[step] = (signed char)(state = bmm[(size_t)i * pitch + j]);
      i_steps[step] = i;
      j_steps[step] = j;
      if (state == MAC_MM) matched++;
      switch (state) {
        case MAC_MM: i--; j--; break;
        case MAC_IM: j--; break;
        case MAC_MI: i--; break;
        case MAC_STOP: break;
        default: state = 0; break;
      }
    }
  }
  states[step] = MAC_MM;
  *nsteps = step;
  *matched_cols = matched;
  float sum = 0.0f;
  for (int s = 1; s <= step; ++s) {
    if (states[s] == MAC_MM) {
      S[s] = hho_fast_log2(hho_dot20_scalar(qp + (size_t)i_steps[s] * 20, tp + (size_t)j_steps[s] * 20));
      P[s] = post[(size_t)i_steps[s] * pitch + j_steps[s]];
      sum += P[s]; /* t.nss_dssp < 0: every aligned pair counts (:224) */
    } else {
      S[s] = P[s] = 0.0f;
    }
  }
  *sum_of_probs = sum;
  return 0;
}

double hho_bench_align(const hho_params *par, const float *qp, const float *qtr, int Lq, int N, const int *L,
                       const float *const *p, const float *const *tr, int threads, float *score, int *i2, int *j2) {
  struct timespec t0, t1;
  if (threads < 1) threads = 1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int k = 0; k < N; k++) {
    hho_align(par, qp, qtr, Lq, p[k], tr[k], L[k], L[k], NULL, NULL, &score[k], &i2[k], &j2[k], NULL);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* N templates with backtrace: hho_align + hho_backtrace + hho_score_for_backtrace per template (single-length batches),
 * OpenMP over templates; the per-template outputs and the two path checksums of ref_bench_hits (oracle/ref_harness.cpp). */
double hho_bench_hits(const hho_params *par, const float *qp, const float *qtr, int Lq, int N, const int *L,
                      const float *const *p, const float *const *tr, int threads, float *score, int *i2, int *j2, int *i1,
                      int *j1, int *nsteps, int *matched_cols, float *hit_score, unsigned long long *path_hash,
                      unsigned long long *s_hash) {
  struct timespec t0, t1;
  if (threads < 1) threads = 1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int k = 0; k < N; k++) {
    const int Lt = L[k], cap = Lq + Lt + 4;
    unsigned char *bt = (unsigned char *)calloc((size_t)(Lq + 1) * (Lt + 1), 1);
    int *is = (int *)calloc(cap, sizeof(int)), *js = (int *)calloc(cap, sizeof(int));
    signed char *st = (signed char *)calloc(cap, 1);
    float *S = (float *)calloc(cap, sizeof(float));
    float sss = 0;
    hho_align(par, qp, qtr, Lq, p[k], tr[k], Lt, Lt, NULL, NULL, &score[k], &i2[k], &j2[k], bt);
    hho_backtrace(bt, Lt + 1, i2[k], j2[k], is, js, st, cap, &nsteps[k], &matched_cols[k]);
    hho_score_for_backtrace(par, qp, p[k], NULL, is, js, st, nsteps[k], score[k], S, &hit_score[k], &sss);
    i1[k] = is[nsteps[k]];
    j1[k] = js[nsteps[k]];
    unsigned long long h = 0, hs = 0;
    for (int s = 1; s <= nsteps[k]; s++) {
      const unsigned long long w = 2ull * (unsigned long long)s + 1ull;
      unsigned int bits;
      h += (((unsigned long long)is[s] * 1000003ull + (unsigned long long)js[s]) * 31ull + (unsigned long long)(unsigned char)st[s]) * w;
      memcpy(&bits, &S[s], 4);
      hs += (unsigned long long)bits * w;
    }
    path_hash[k] = h;
    s_hash[k] = hs;
    free(bt);
    free(is);
    free(js);
    free(st);
    free(S);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
