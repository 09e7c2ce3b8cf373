// hhposteriordecoderrunner_hip.cpp -- DROP-IN replacement for src/hhposteriordecoderrunner.cpp of hh-suite v3.3.0.
//
// Same class, same constructor, same executeComputation(q, hits, par, qsc, pb, S, Sim, R) as declared by the reference's
// own src/hhposteriordecoderrunner.h (untouched); a maintainer compiles this file instead of the reference's and links
// libhhviterbi_hip.so.  HHblits::perform_realign (src/hhblits.cpp:973-1034) keeps calling it and finds in every Hit what
// PosteriorDecoder::realign leaves there (src/hhposteriordecoder.cpp:86-119, src/hhbacktracemac.cpp:113-240):
//     i, j, states, S, S_ss, P_posterior [nsteps+1], alt_i / alt_j (emptied at the end, :108-113), i1, j1, i2, j2, nsteps, matched_cols, sum_of_probs,
//     Pforward, state = STOP, min_overlap = 0, realign_around_viterbi = true;
//     score, score_ss, score_aass, P-values, E-value, Probab stay the Viterbi ones (memorizeHitValues / restoreHitValues).
// Forward, backward, posterior, MAC DP and MAC backtrace run on the GPU (hhv_mac_realign_hits: one wavefront per hit,
// the cell-off masks built on the device), bit-exact against the reference's doubles (tests/test_mac.py).
//
// Control flow of the reference kept here: the hits are grouped by template NAME and ordered by irep inside a group
// (:52-66); the r-th alignment of a template excludes the cells (+-2) of the MAC alignments 1..r-1 of the same template
// (alignment_to_exclude, :104-106); templates are independent.  The reference runs the groups in an OpenMP loop, one hit
// after the other inside a group; here ROUND r realigns the r-th hit of every group in one launch.  The template of a
// group comes from the resident template cache of the Viterbi stage when it is there and its reading did not depend on the
// sequence weighting (an .hhm text, or par.wg = 1: prepared on the device for this query and fetched back in one copy - no
// parsing); otherwise it is read and prepared once with the reference's own code
// (getTemplateHMM + PrepareTemplateHMM with linear transitions, :98-99), in parallel over the groups.
//
// The sparse forward / backward / posterior lists and the two profiles of writeProfilesToHits (hit.forward_matrix, ...;
// src/hhbacktracemac.cpp:14-110) are read by HitList::PrintMatrices only (the hidden -o_matrices output): they are produced
// when that output is asked for (par.matrices_output_file; HHV_MAC_LISTS=1 forces them) - entry for entry the reference's
// (hhv_mac_list) - and left NULL otherwise (the reference computes them for every hit of every search; PrintMatrices skips
// hits without them).  Secondary-structure scoring inside forward / backward (hit.ssm2 = 1 or 2: a query
// with predicted SS against templates with DSSP records - the HHpred case - or the reverse; ssm2 = 3 scores nothing in the
// reference, Viterbi::ScoreSS has no case 3) runs on the device from tables of fpow2(ScoreSS) (hhv_mac_set_ss).  One cell
// column is not reproducible: for column 1 the reference calls ScoreSS with a stale loop variable and reads the template's
// SS state one element past its last column (src/hhforwardalgorithm.cpp:77) - zero in a fresh HMM, which is what is used
// here.  Templates of any length: the hits of a round are launched by length class on overlapping streams (template and row
// state in LDS up to about 800 columns, row state in LDS up to 2046, row state in global memory beyond).  Not supported (the run stops with a
// message instead of computing something else): self alignments (hit.self; nothing in v3.3.0 sets it, src/hhhit.cpp:13).
#include <sys/time.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "hhposteriordecoderrunner.h"
#include "hhviterbi_hip.h"
#include "hhv_template_cache.h"

#ifdef OPENMP
#include <omp.h>
#endif

namespace {

double mac_now() {
  struct timeval tv;
  gettimeofday(&tv, NULL);
  return tv.tv_sec + 1e-6 * tv.tv_usec;
}

void mac_check(int rc, const char* what) {
  if (rc == HHV_OK) return;
  HH_LOG(ERROR) << "hhviterbi_hip: " << what << " failed: " << hhv_last_error() << std::endl;
  exit(rc == HHV_E_MEMORY ? 3 : 4);
}

// std::sort predicate of the reference (src/hhposteriordecoderrunner.cpp:27-31), on the pointers it is applied to
bool irep_less(const Hit* a, const Hit* b) { return a->irep < b->irep; }

std::vector<int32_t> mac_region_pairs(char* exclstr) {  // exclude_regions, src/hhposteriordecoder.cpp:121-149
  std::vector<int32_t> out;
  if (!exclstr) return out;
  char* ptr = exclstr;
  while (true) {
    int a = abs(strint(ptr));
    int b = abs(strint(ptr));
    if (!ptr) break;
    out.push_back(a < 1 ? 1 : a);
    out.push_back(b);
  }
  return out;
}

// writeProfilesToHits (src/hhbacktracemac.cpp:14-110): the three lists as arrays of float[3] = {i, j, value} owned by the Hit
// (freed by Hit::Delete like the reference's), the profiles = the sums of a list's values per query row, in list order
void mac_lists_to_hit(hhv_macset* ms, int b, int Lq, Hit& hit) {
  if (hit.forward_profile) delete[] hit.forward_profile;
  hit.forward_profile = new float[Lq + 1];
  if (hit.backward_profile) delete[] hit.backward_profile;
  hit.backward_profile = new float[Lq + 1];
  for (int i = 0; i <= Lq; i++) hit.backward_profile[i] = hit.forward_profile[i] = 0;
  float*** const mat[3] = {&hit.backward_matrix, &hit.forward_matrix, &hit.posterior_matrix};
  size_t* const cnt[3] = {&hit.backward_entries, &hit.forward_entries, &hit.posterior_entries};
  float* const prof[3] = {hit.backward_profile, hit.forward_profile, NULL};
  const int which[3] = {1, 0, 2};  // (the reference fills backward first)
  std::vector<int32_t> li, lj;
  std::vector<float> lv;
  for (int w = 0; w < 3; ++w) {
    if (*mat[w]) {
      for (size_t e = 0; e < *cnt[w]; e++) delete[] (*mat[w])[e];
      delete[] *mat[w];
    }
    int64_t n = hhv_mac_list(ms, b, which[w], 0, NULL, NULL, NULL);
    if (n < 0) mac_check((int)n, "hhv_mac_list");
    li.resize((size_t)n);
    lj.resize((size_t)n);
    lv.resize((size_t)n);
    if (n > 0) {
      const int64_t m = hhv_mac_list(ms, b, which[w], n, li.data(), lj.data(), lv.data());
      if (m != n) mac_check(m < 0 ? (int)m : HHV_E_STATE, "hhv_mac_list");
    }
    *cnt[w] = (size_t)n;
    *mat[w] = new float*[(size_t)n];
    for (int64_t e = 0; e < n; ++e) {
      float* t = new float[3];
      t[0] = li[(size_t)e];
      t[1] = lj[(size_t)e];
      t[2] = lv[(size_t)e];
      (*mat[w])[e] = t;
      if (prof[w]) prof[w][li[(size_t)e]] += lv[(size_t)e];
    }
  }
}

// The device context is the process-wide one of the resident template cache (hhv_template_cache.h): stream, tables and the
// recycled device block of the MAC stage survive from call to call; concurrent callers (hhblits_omp) take turns on it.

struct PreparedTemplate {
  int L;
  std::vector<float> p, tr_lin;  // [(L+1)*20], [(L+1)*7] linear transitions with the boundary values of :159-167
  int has_dssp;
  std::vector<char> dssp, pred, conf;  // t.ss_dssp (sum_of_probs, src/hhbacktracemac.cpp:196-197), ss_pred / ss_conf (S_ss)
};

}  // namespace

PosteriorDecoderRunner::PosteriorDecoderRunner(PosteriorMatrix** posterior_matrices, ViterbiMatrix** backtrace_matrix,
                                               const int n_threads, const float ssw,
                                               const float S73[NDSSP][NSSPRED][MAXCF],
                                               const float S33[NSSPRED][MAXCF][NSSPRED][MAXCF],
                                               const float S37[NSSPRED][MAXCF][NDSSP])
    : S73(S73), S33(S33), S37(S37), m_posterior_matrices(posterior_matrices), m_backtrace_matrix(backtrace_matrix),
      m_n_threads(n_threads) {}

PosteriorDecoderRunner::~PosteriorDecoderRunner() {}

// src/hhposteriordecoderrunner.cpp:146-155
void PosteriorDecoderRunner::initializeQueryHMMTransitions(HMM& q) {
  q.tr[0][M2D] = q.tr[0][M2I] = 0.0f;
  q.tr[0][I2M] = q.tr[0][I2I] = 0.0f;
  q.tr[0][D2M] = q.tr[0][D2D] = 0.0f;
  q.tr[q.L][M2M] = 1.0f;
  q.tr[q.L][M2D] = q.tr[q.L][M2I] = 0.0f;
  q.tr[q.L][I2M] = q.tr[q.L][I2I] = 0.0f;
  q.tr[q.L][D2M] = 1.0f;
  q.tr[q.L][D2D] = 0.0f;
}

void PosteriorDecoderRunner::executeComputation(HMM& q, std::vector<Hit*> hits, Parameters& par, const float qsc, float* pb,
                                                const float S[20][20], const float Sim[20][20], const float R[20][20]) {
  const bool timing = getenv("HHV_DROPIN_TIMING") != NULL;  // prints where the wall time goes (stderr)
  double t_mark = mac_now(), t_read = 0, t_create = 0, t_device = 0, t_hits = 0;
  HMM* q_hmm = &q;
  q_hmm->Log2LinTransitionProbs(1.0);       // :48
  initializeQueryHMMTransitions(*q_hmm);    // :50
  if (hits.empty()) return;

  // ---- groups: one per template name, ordered by irep (:52-66) ----
  std::map<std::string, std::vector<Hit*> > alignments_map;
  for (size_t i = 0; i < hits.size(); i++) {
    Hit* h = hits[i];
    if (h->self) {
      HH_LOG(ERROR) << "hhviterbi_hip: MAC realignment of self alignments is not supported on the device" << std::endl;
      exit(4);
    }
    alignments_map[h->entry->getName()].push_back(h);
  }
  std::vector<std::vector<Hit*> > alignment;
  size_t rounds = 0;
  for (std::map<std::string, std::vector<Hit*> >::iterator it = alignments_map.begin(); it != alignments_map.end(); ++it) {
    std::sort(it->second.begin(), it->second.end(), irep_less);
    alignment.push_back(it->second);
    rounds = std::max(rounds, it->second.size());
  }
  const int n_groups = (int)alignment.size();

  // ---- the template of every group ----
  // Templates the Viterbi stage left in the resident cache (raw columns on the device) are prepared there for this query and
  // fetched back prepared - nothing is parsed; the others are read and prepared by the reference's code (:98-99).  Either way
  // the transitions become linear on the host: 2^x is the host's powf (HMM::Log2LinTransitionProbs, src/hhhmm.cpp:2305-2313).
  hhv_dropin::TemplateCache& tc = hhv_dropin::cache();
  const int threads = m_n_threads > 0 ? m_n_threads : 1;
  std::vector<PreparedTemplate> tmpl(n_groups);
  std::vector<const hhv_dropin::CachedTemplate*> cached(n_groups, (const hhv_dropin::CachedTemplate*)NULL);
  // The cache holds what ViterbiRunner read with use_global_weights = 1 (src/hhviterbirunner.cpp:143); this stage reads with
  // par.wg (:98), and for templates built from alignments the sequence weighting changes the HMM - so the cache stands in
  // for the reader when par.wg asks for the same weighting (-wg), or for entries that are .hhm texts (weights_free).
  bool use_cache = tc.enabled && hhv_dropin::device_prepare_covers(par);
  {
    std::lock_guard<std::mutex> lock(tc.device);
    use_cache = use_cache && !tc.map.empty() && tc.nseqdis == par.nseqdis && tc.ssm == par.ssm;
    if (use_cache) {
      tc.active++;  // keeps the entries alive until the end of this call
      for (int g = 0; g < n_groups; ++g) {
        std::unordered_map<std::string, hhv_dropin::CachedTemplate>::const_iterator it =
            tc.map.find(hhv_dropin::cache_key(alignment[g][0]->entry));
        // (templates a multi-device search parked on another device are read again here, like uncached ones)
        if (it != tc.map.end() && it->second.dev == 0 && (par.wg == 1 || it->second.weights_free)) cached[g] = &it->second;
      }
    }
  }
  std::vector<int> to_read;
  for (int g = 0; g < n_groups; ++g)
    if (!cached[g]) to_read.push_back(g);
  std::vector<HMM*> t_hmm(threads, (HMM*)NULL);
  const int n_read = (int)to_read.size();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int r = 0; r < n_read; ++r) {
    const int g = to_read[r];
    int tid = 0;
#ifdef OPENMP
    tid = omp_get_thread_num();
#endif
    if (!t_hmm[tid]) t_hmm[tid] = new HMM(MAXSEQDIS, par.maxres);
    HMM* t = t_hmm[tid];
    int format_tmp = 0;
    alignment[g][0]->entry->getTemplateHMM(par, par.wg, qsc, format_tmp, pb, S, Sim, t);
    PrepareTemplateHMM(par, q_hmm, t, format_tmp, true, pb, R);
    PreparedTemplate& pt = tmpl[g];
    pt.L = t->L;
    pt.p.assign((size_t)(t->L + 1) * 20, 0.0f);
    pt.tr_lin.resize((size_t)(t->L + 1) * 7);
    for (int i = 0; i <= t->L; ++i) {
      if (i >= 1) memcpy(&pt.p[(size_t)i * 20], t->p[i], 20 * sizeof(float));
      memcpy(&pt.tr_lin[(size_t)i * 7], t->tr[i], 7 * sizeof(float));
    }
    pt.has_dssp = t->nss_dssp >= 0;
    pt.dssp.assign(t->ss_dssp, t->ss_dssp + t->L + 1);
    pt.pred.assign(t->ss_pred, t->ss_pred + t->L + 1);
    pt.conf.assign(t->ss_conf, t->ss_conf + t->L + 1);
  }
  for (int k = 0; k < threads; ++k) delete t_hmm[k];
  t_read = mac_now() - t_mark;
  t_mark = mac_now();

  std::lock_guard<std::mutex> device_lock(tc.device);
  mac_check(hhv_dropin::ensure_context(tc), "hhv_create");
  hhv_ctx* ctx = tc.slots[0].ctx;  // the realign stage works on the primary device
  if (n_read < n_groups) {
    // PrepareTemplateHMM on the device (hhv_prepare_subset), one launch per raw set, and the prepared records back in one copy
    // against the background the cached templates were read with (their files' NULL line, hhv_template_cache.h null_pb) - the
    // reference reads every template again here and HMM::Read puts that line into pb right before PrepareTemplateHMM (:98-99); the
    // caller's pb may hold the COMPO line of an HMMER-format template the Viterbi stage read last
    const hhv_prep_params prep = hhv_dropin::prepare_params(par, tc.null_pb_set ? tc.null_pb : pb, R);
    std::map<hhv_rawset*, std::vector<int> > by_raw;
    for (int g = 0; g < n_groups; ++g)
      if (cached[g]) by_raw[cached[g]->raw].push_back(g);
    for (std::map<hhv_rawset*, std::vector<int> >::iterator b = by_raw.begin(); b != by_raw.end(); ++b) {
      const std::vector<int>& mem = b->second;
      std::vector<int32_t> ids(mem.size());
      for (size_t x = 0; x < mem.size(); ++x) ids[x] = cached[mem[x]]->index;
      hhv_tset* set = NULL;
      mac_check(hhv_prepare_subset(ctx, b->first, &prep, q.pav, ids.data(), (int32_t)ids.size(), &set), "hhv_prepare_subset");
      std::vector<float> rec((size_t)hhv_tset_records(set) * 28);
      mac_check(hhv_tset_download(ctx, set, rec.data()), "hhv_tset_download");
      hhv_tset_free(set);
      size_t at = 0;  // header record of the template
      for (size_t x = 0; x < mem.size(); ++x) {
        const hhv_dropin::CachedTemplate& ct = *cached[mem[x]];
        PreparedTemplate& pt = tmpl[mem[x]];
        const int L = ct.L;
        pt.L = L;
        pt.p.assign((size_t)(L + 1) * 20, 0.0f);
        pt.tr_lin.assign((size_t)(L + 1) * 7, 0.0f);
        // column record j (layout: DESIGN.md section 2): [0..19] p[j], [20..24] tr[j-1][M2M, M2D, D2M, D2D, I2M], [25..26] tr[j][I2I, M2I]
        for (int j = 1; j <= L; ++j) {
          const float* w = &rec[(at + j) * 28];
          memcpy(&pt.p[(size_t)j * 20], w, 20 * sizeof(float));
          float* prev = &pt.tr_lin[(size_t)(j - 1) * 7];
          prev[M2M] = w[20];
          prev[M2D] = w[21];
          prev[D2M] = w[22];
          prev[D2D] = w[23];
          prev[I2M] = w[24];
          float* cur = &pt.tr_lin[(size_t)j * 7];
          cur[I2I] = w[25];
          cur[M2I] = w[26];
        }
        at += (size_t)L + 1;
        pt.has_dssp = !ct.ss.dssp.empty();
        pt.dssp.assign(L + 1, 0);
        pt.pred.assign(L + 1, 0);
        pt.conf.assign(L + 1, 0);
        for (int j = 1; j <= L; ++j) {
          if (!ct.ss.dssp.empty()) pt.dssp[j] = (char)ct.ss.dssp[j];
          if (!ct.ss.pred.empty()) pt.pred[j] = (char)ct.ss.pred[j];
          if (!ct.ss.conf.empty()) pt.conf[j] = (char)ct.ss.conf[j];
        }
      }
    }
    // HMM::Log2LinTransitionProbs(1.0) of the rows that keep their value (rows 0 and L are assigned below)
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int g = 0; g < n_groups; ++g) {
      if (!cached[g]) continue;
      PreparedTemplate& pt = tmpl[g];
      for (size_t e = 7; e < (size_t)pt.L * 7; ++e) pt.tr_lin[e] = powf(2.0f, 1.0f * pt.tr_lin[e]);
    }
  }
  // initializeForAlignment (src/hhposteriordecoder.cpp:159-167)
  for (int g = 0; g < n_groups; ++g) {
    PreparedTemplate& pt = tmpl[g];
    float* t0 = &pt.tr_lin[0];
    float* tL = &pt.tr_lin[(size_t)pt.L * 7];
    t0[M2M] = 1.0f;
    t0[M2D] = t0[M2I] = 0.0f;
    t0[I2M] = t0[I2I] = 0.0f;
    t0[D2M] = t0[D2D] = 0.0f;
    tL[M2M] = 1.0f;
    tL[M2D] = tL[M2I] = 0.0f;
    tL[I2M] = tL[I2I] = 0.0f;
    tL[D2M] = 1.0f;
    tL[D2D] = 0.0f;
  }
  t_create = mac_now() - t_mark;

  // ---- the query as the device wants it ----
  std::vector<float> q_p((size_t)(q.L + 1) * 20, 0.0f), q_tr((size_t)(q.L + 1) * 7);
  for (int i = 0; i <= q.L; ++i) {
    if (i >= 1) memcpy(&q_p[(size_t)i * 20], q.p[i], 20 * sizeof(float));
    memcpy(&q_tr[(size_t)i * 7], q.tr[i], 7 * sizeof(float));
  }
  std::vector<int32_t> q_ranges = mac_region_pairs(par.exclstr), t_ranges = mac_region_pairs(par.template_exclstr);

  // ---- secondary-structure scoring inside forward / backward (hit.ssm2 = 1 or 2): the factors fpow2(ScoreSS) as tables ----
  // Viterbi::ScoreSS(q, t, i, j, ssw, hit.ssm2, ...) (src/hhviterbi.h:193-211) is ssw * S37[q_pred][q_conf][t_dssp] for
  // ssm2 = 1 (HMM::PRED_DSSP) and ssw * S73[q_dssp][t_pred][t_conf] for 2 (HMM::DSSP_PRED); 3 has no case there (factor 1).
  bool any_ss = false;
  for (size_t i = 0; i < hits.size(); i++) any_ss = any_ss || hits[i]->ssm2 == 1 || hits[i]->ssm2 == 2;
  std::vector<float> ss_tables;
  std::vector<uint8_t> ss_qidx;
  if (any_ss) {
    ss_tables.assign(2 * 352, 1.0f);
    for (int qp = 0; qp < 44; ++qp)
      for (int d = 0; d < 8; ++d) ss_tables[qp * 8 + d] = fpow2(par.ssw * S37[qp / MAXCF][qp % MAXCF][d]);
    for (int qd = 0; qd < 8; ++qd)
      for (int tp = 0; tp < 44; ++tp) ss_tables[352 + qd * 44 + tp] = fpow2(par.ssw * S73[qd][tp / MAXCF][tp % MAXCF]);
    ss_qidx.assign((size_t)2 * (q.L + 2), 0);
    for (int i = 1; i <= q.L; ++i) {
      if (q.nss_pred >= 0) ss_qidx[i] = (uint8_t)(q.ss_pred[i] * MAXCF + q.ss_conf[i]);
      if (q.nss_dssp >= 0) ss_qidx[(size_t)(q.L + 2) + i] = (uint8_t)q.ss_dssp[i];
    }
  }


  static const bool force_lists = getenv("HHV_MAC_LISTS") && atoi(getenv("HHV_MAC_LISTS")) != 0;
  const bool want_lists = force_lists || par.matrices_output_file[0] != 0;
  // ---- round r: the r-th alignment of every template ----
  for (size_t r = 0; r < rounds; ++r) {
    std::vector<int> group_of;
    for (int g = 0; g < n_groups; ++g)
      if (alignment[g].size() > r) group_of.push_back(g);
    const int n = (int)group_of.size();
    std::vector<hhv_mac_input> in(n);
    std::vector<std::vector<int32_t> > path_i(n), path_j(n), ex_i(n), ex_j(n);
    std::vector<const float*> tp(n), ttr(n);
    std::vector<int32_t> Lt(n);
    for (int b = 0; b < n; ++b) {
      const int g = group_of[b];
      Hit* hit = alignment[g][r];
      path_i[b].assign(hit->i, hit->i + hit->nsteps + 1);
      path_j[b].assign(hit->j, hit->j + hit->nsteps + 1);
      for (size_t e = 0; e < r; ++e) {  // alignment_to_exclude: alt_i / alt_j of the earlier alignments (:104-106)
        const Hit* prev = alignment[g][e];
        ex_i[b].insert(ex_i[b].end(), prev->alt_i->begin(), prev->alt_i->end());
        ex_j[b].insert(ex_j[b].end(), prev->alt_j->begin(), prev->alt_j->end());
      }
      in[b].i1 = hit->i1;
      in[b].j1 = hit->j1;
      in[b].i2 = hit->i2;
      in[b].j2 = hit->j2;
      in[b].nsteps = hit->nsteps;
      in[b].i = path_i[b].data();
      in[b].j = path_j[b].data();
      in[b].n_excluded = (int32_t)ex_i[b].size();
      in[b].excluded_i = ex_i[b].data();
      in[b].excluded_j = ex_j[b].data();
      tp[b] = tmpl[g].p.data();
      ttr[b] = tmpl[g].tr_lin.data();
      Lt[b] = tmpl[g].L;
    }
    if (any_ss) {
      std::vector<int32_t> mode(n, 0);
      std::vector<std::vector<uint8_t> > tidx(n);
      std::vector<const uint8_t*> tidx_p(n, (const uint8_t*)NULL);
      bool round_has_ss = false;
      for (int b = 0; b < n; ++b) {
        const Hit* hit = alignment[group_of[b]][r];
        const PreparedTemplate& pt = tmpl[group_of[b]];
        if (hit->ssm2 != 1 && hit->ssm2 != 2) continue;
        mode[b] = hit->ssm2;
        round_has_ss = true;
        // entry L+1 stays 0: the reference reads the element past the template's records there (the stale loop variable of
        // src/hhforwardalgorithm.cpp:77), which is zero in a freshly allocated HMM
        tidx[b].assign((size_t)pt.L + 2, 0);
        for (int j = 1; j <= pt.L; ++j)
          tidx[b][j] = hit->ssm2 == 1 ? (uint8_t)pt.dssp[j] : (uint8_t)(pt.pred[j] * MAXCF + pt.conf[j]);
        tidx_p[b] = tidx[b].data();
      }
      if (round_has_ss)
        mac_check(hhv_mac_set_ss(ctx, ss_tables.data(), ss_qidx.data(), q.L, n, mode.data(), tidx_p.data(), Lt.data()), "hhv_mac_set_ss");
    }
    hhv_macset* ms = NULL;
    std::vector<hhv_mac_hit> res(n);
    t_mark = mac_now();
    mac_check(hhv_mac_set_lists(ctx, want_lists ? 1 : 0), "hhv_mac_set_lists");
    mac_check(hhv_mac_realign_hits(ctx, q_p.data(), q_tr.data(), q.L, n, Lt.data(), tp.data(), ttr.data(), in.data(),
                                   (int32_t)q_ranges.size() / 2, q_ranges.data(), (int32_t)t_ranges.size() / 2, t_ranges.data(),
                                   par.loc, par.shift, par.mact, &ms, res.data()),
              "hhv_mac_realign_hits");
    t_device += mac_now() - t_mark;
    t_mark = mac_now();
    for (int b = 0; b < n; ++b) {
      const int g = group_of[b];
      Hit& hit = *alignment[g][r];
      const hhv_mac_hit& m = res[b];
      // initializeForAlignment (:168-177) and initializeBacktrace (src/hhbacktracemac.cpp:273-304)
      if (hit.alt_i) delete hit.alt_i;
      hit.alt_i = new std::vector<int>();
      if (hit.alt_j) delete hit.alt_j;
      hit.alt_j = new std::vector<int>();
      hit.realign_around_viterbi = true;
      hit.min_overlap = 0;  // src/hhmacalgorithm.cpp:52
      hit.Pforward = m.Pforward;
      hit.i2 = m.i2;
      hit.j2 = m.j2;
      if (hit.i) delete[] hit.i;
      if (hit.j) delete[] hit.j;
      if (hit.states) delete[] hit.states;
      if (hit.S) delete[] hit.S;
      if (hit.S_ss) delete[] hit.S_ss;
      if (hit.P_posterior) delete[] hit.P_posterior;
      const int cap = std::max(m.i2 + m.j2 + 2, m.nsteps + 1);
      hit.i = new int[cap];
      hit.j = new int[cap];
      hit.states = new char[cap];
      hit.S = new float[m.nsteps + 1];
      hit.S_ss = new float[m.nsteps + 1];
      hit.P_posterior = new float[m.nsteps + 1];
      int32_t ns = 0;
      mac_check(hhv_mac_path(ms, b, m.nsteps + 1, hit.i, hit.j, (int8_t*)hit.states, hit.S, hit.P_posterior, &ns), "hhv_mac_path");
      // backtraceMAC (:113-240)
      hit.nsteps = m.nsteps;
      hit.matched_cols = m.matched_cols;
      hit.i1 = m.i1;
      hit.j1 = m.j1;
      hit.state = ViterbiMatrix::STOP;
      if (m.nsteps == 0) {  // the backtrace did not start in a match-match state (:128-138): one cell, no steps
        hit.i[0] = m.i2;
        hit.j[0] = m.j2;
        hit.alt_i->push_back(m.i2);
        hit.alt_j->push_back(m.j2);
        hit.states[0] = ViterbiMatrix::MM;  // hit.states[step] = MM with step = 0 (:164)
      }
      // S_ss[step] = Viterbi::ScoreSS(q, t, i, j, ssw, ssm1 + ssm2, ...) for match states (:188-189); with ssm2 in {0, 3}
      // the sum is 0 or 3 (no such case in ScoreSS: 0) unless SS is scored AFTER the alignment (ssm1 = 1 or 2)
      const int ssm = hit.ssm1 + hit.ssm2;
      for (int step = 1; step <= m.nsteps; ++step) {
        hit.alt_i->push_back(hit.i[step]);
        hit.alt_j->push_back(hit.j[step]);
        float s_ss = 0.0f;
        if (hit.states[step] == ViterbiMatrix::MM) {
          const int i = hit.i[step], j = hit.j[step];
          if (ssm == HMM::PRED_DSSP) s_ss = par.ssw * S37[(int)q.ss_pred[i]][(int)q.ss_conf[i]][(int)tmpl[g].dssp[j]];
          else if (ssm == HMM::DSSP_PRED) s_ss = par.ssw * S73[(int)q.ss_dssp[i]][(int)tmpl[g].pred[j]][(int)tmpl[g].conf[j]];
          else if (ssm == HMM::PRED_PRED)
            s_ss = par.ssw * S33[(int)q.ss_pred[i]][(int)q.ss_conf[i]][(int)tmpl[g].pred[j]][(int)tmpl[g].conf[j]];
        }
        hit.S_ss[step] = s_ss;
      }
      hit.sum_of_probs = m.sum_of_probs;
      if (tmpl[g].has_dssp) {  // only columns with a resolved DSSP state count (:196-197)
        float sum = 0.0;
        for (int step = 1; step <= m.nsteps; ++step)
          if (hit.states[step] == ViterbiMatrix::MM && tmpl[g].dssp[hit.j[step]] > 0) sum += hit.P_posterior[step];
        hit.sum_of_probs = sum;
      }
      // score, score_ss, score_aass, Pval, Pvalt, logPval, logPvalt, Eval, logEval, Probab: untouched = restoreHitValues
      if (want_lists) mac_lists_to_hit(ms, b, q.L, hit);  // writeProfilesToHits (:117)
    }
    hhv_macset_free(ms);
    t_hits += mac_now() - t_mark;
  }
  if (timing)
    fprintf(stderr, "hhposteriordecoderrunner_hip: %zu hits of %d templates (%d read) in %zu rounds; read+prepare %.3f s, device prepare %.3f s, "
            "device (staging, masks, forward/backward/MAC, paths) %.3f s, Hit objects %.3f s\n", hits.size(), n_groups, n_read, rounds,
            t_read, t_create, t_device, t_hits);
  // "clear all backtrace paths" (:108-113): the vectors stay allocated, empty
  for (size_t i = 0; i < hits.size(); i++) {
    hits[i]->alt_i->clear();
    hits[i]->alt_j->clear();
  }
  // the caller's pb as the reference leaves it with one thread: the background of the template it read last - the last group (a
  // cached one is not read here; a group that was read has left its own)
  if (n_groups > 0 && cached[n_groups - 1] && tc.null_pb_set) memcpy(pb, tc.null_pb, sizeof(tc.null_pb));
  if (use_cache) tc.active--;  // (the device lock is still held)
}
