// hhviterbirunner_hip.cpp -- DROP-IN replacement for src/hhviterbirunner.cpp of hh-suite v3.3.0.
//
// A maintainer compiles THIS file instead of src/hhviterbirunner.cpp (same class, same signature, declared by
// the reference's own src/hhviterbirunner.h, which stays untouched) and links libhhviterbi_hip.so.  Every caller
// (HHblits::run x3, RescoreWithViterbiKeepAlignment, HHalign::run) keeps calling
//     ViterbiRunner::alignment(par, q_simd, dbfiles, qsc, pb, S, Sim, R, ssm_mode, S73, S33, S37)
// and receives the same std::vector<Hit> (same fields, same ownership: the path arrays are new[]'d here and freed
// by Hit::Delete).  The host part the reference runs per template - HHEntry::getTemplateHMM + PrepareTemplateHMM
// (src/hhviterbirunner.cpp:144-147) - is still the reference's own code, called from here under OpenMP; what
// moves to the GPU is everything behind Viterbi::Align / Backtrace / ScoreForBacktrace (:24-31) and the
// exclusion masks (:152-164), through the C ABI of include/hhviterbi_hip.h.  No DP arithmetic in this file.
//
// What is kept from the reference's control flow (src/hhviterbirunner.cpp:75-210):
//   * alternative-alignment rounds 0..par.altali-1, work list of round r+1 = entries of the hits with
//     score > par.smin in merge order (:104,260-268), irep = round + 1 (:257)
//   * blocks of 2000 templates in round 0 when par.early_stopping_filter (:109-111), each block sorted by length
//     with the same std::sort call (:117-119), early stop = calculateEarlyStop < block * par.filter_thresh (:178-188)
//   * SIMD batches of VECSIZE_FLOAT consecutive templates of the sorted block decide the secondary-structure mode
//     (consensus of HMM::computeScoreSSMode over the batch, then the selection chain of :14-22)
//   * exclusion of earlier alignments keyed by the template NAME (:262-268,273-289), -excl / -template_excl (:157-164)
//   * Hit fields of ViterbiConsumerThread::align (:35-62)
// Differences, by design:
//   * a template is read and prepared ONCE; it stays resident on the device for the later rounds (the reference
//     reads and prepares it again in every round), and the Hit of a later round copies the template information of
//     its first-round Hit instead of calling initHitFromHMM on a freshly read HMM
//   * hits come back in the order of the sorted block (= the reference's order with one thread; with several
//     threads the reference's order depends on the OpenMP schedule)
//   * global mode (par.loc = 0): every template is maximised over its OWN last column; the reference maximises over
//     the last column of the longest template of the SIMD batch (SURVEY.md 8a, row A1 "batch-composition quirk"),
//     so results differ for the shorter templates of mixed-length batches.  Local mode (the default) is unaffected.
//   * the device is chosen with the environment variable HHV_DEVICE (default 0)
// Errors follow the reference's convention at this layer: HH_LOG(ERROR) + exit(code) (src/hhsearch.h:6).
#include <map>
#include <string>
#include <vector>

#include "hhviterbirunner.h"
#include "hhviterbi_hip.h"

#ifdef OPENMP
#include <omp.h>
#endif

namespace {

void hip_check(int rc, const char* what) {
  if (rc == HHV_OK) return;
  HH_LOG(ERROR) << "hhviterbi_hip: " << what << " failed: " << hhv_last_error() << std::endl;
  exit(rc == HHV_E_MEMORY ? 3 : 4);
}

// a template of the search: resident on the device after its first alignment
struct ResidentTemplate {
  hhv_tset* set;       // the chunk it was uploaded with
  int32_t index;       // its index inside that set
  int L;
  int ss_pair_mode;    // HMM::computeScoreSSMode(q, t)
  size_t first_hit;    // index of its first-round Hit in the result vector (source of the template information)
  std::vector<int8_t> ss_pred, ss_conf, ss_dssp;  // [L+1], empty when the template has no such record
};

// host copy of one prepared template on its way to the device
struct Prepared {
  std::vector<float> p, tr;
  std::vector<int8_t> ss_pred, ss_conf, ss_dssp;
};

const int kCacheToEnum[7] = {4 /*I2I*/, 1 /*M2I*/, 0 /*M2M*/, 2 /*M2D*/, 5 /*D2M*/, 6 /*D2D*/, 3 /*I2M*/};

// Lane 0 of an HMMSimd that holds ONE HMM (HMMSimd::MapHMMVector, src/hhhmmsimd.cpp:86-160) -> the prepared-profile
// layout of the C ABI: p[(L+1)*20], tr[(L+1)*7] in the enum order of src/hhdecl.h:68.  HMM::tr and the ss arrays
// are private to HMM; HMMSimd is its friend and publishes them, which is how the reference's kernel sees them too.
void lane0_to_profile(const HMMSimd* s, const HMM* h, Prepared* out) {
  const int L = h->L;
  out->p.resize((size_t)(L + 1) * 20);
  out->tr.resize((size_t)(L + 1) * 7);
  const float* tr_scalar = (const float*)s->tr;
  for (int i = 0; i <= L; ++i) {
    for (int a = 0; a < 20; ++a) out->p[(size_t)i * 20 + a] = s->p[i][a * VECSIZE_FLOAT];
    for (int c = 0; c < 7; ++c) out->tr[(size_t)i * 7 + kCacheToEnum[c]] = tr_scalar[((size_t)i * 7 + c) * VECSIZE_FLOAT];
  }
  for (int a = 0; a < 20; ++a) out->p[a] = 0.0f;  // row 0 is never read by the DP
  out->ss_pred.clear();
  out->ss_conf.clear();
  out->ss_dssp.clear();
  if (h->nss_pred >= 0) {  // pred_index = ss_pred * MAXCF + ss_conf (:133)
    out->ss_pred.assign(L + 1, 0);
    out->ss_conf.assign(L + 1, 0);
    for (int i = 1; i <= L; ++i) {
      const unsigned v = s->pred_index[(size_t)(i - 1) * VECSIZE_FLOAT];
      out->ss_pred[i] = (int8_t)(v / MAXCF);
      out->ss_conf[i] = (int8_t)(v % MAXCF);
    }
  }
  if (h->nss_dssp >= 0) {
    out->ss_dssp.assign(L + 1, 0);
    for (int i = 1; i <= L; ++i) out->ss_dssp[i] = (int8_t)s->dssp_index[(size_t)(i - 1) * VECSIZE_FLOAT];
  }
}

// the selection chain of ViterbiConsumerThread::align (src/hhviterbirunner.cpp:19-22), literally
int select_ss_mode(int consensus) {
  int m = (consensus & HMM::PRED_DSSP);
  m = (m == 0) ? consensus & HMM::DSSP_PRED : 0;
  m = (m == 0) ? consensus & HMM::PRED_PRED : 0;
  return m;
}

// -excl / -template_excl: the (lo, hi) pairs exclude_regions iterates over (src/hhviterbirunner.cpp:291-329)
std::vector<int32_t> region_pairs(char* exclstr) {
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

// A Hit for a later round of a template whose information the first-round Hit already carries: every owned
// array is duplicated (Hit::Delete frees them per Hit, src/hhhit.cpp:38-62)
void copy_template_info(const Hit& src, Hit* dst) {
  *dst = src;
  dst->longname = new char[strlen(src.longname) + 1];
  strcpy(dst->longname, src.longname);
  dst->name = new char[strlen(src.name) + 1];
  strcpy(dst->name, src.name);
  dst->file = new char[strlen(src.file) + 1];
  strcpy(dst->file, src.file);
  dst->sname = new char*[src.n_display];
  dst->seq = new char*[src.n_display];
  for (int k = 0; k < src.n_display; ++k) {
    dst->sname[k] = new char[strlen(src.sname[k]) + 1];
    strcpy(dst->sname[k], src.sname[k]);
    dst->seq[k] = new char[strlen(src.seq[k]) + 1];
    strcpy(dst->seq[k], src.seq[k]);
  }
  dst->i = dst->j = NULL;
  dst->states = NULL;
  dst->S = dst->S_ss = dst->P_posterior = NULL;
  dst->alt_i = dst->alt_j = NULL;
}

struct SsTables {
  const float (*S73)[NSSPRED][MAXCF];
  const float (*S33)[MAXCF][NSSPRED][MAXCF];
  const float (*S37)[MAXCF][NDSSP];
};

// Viterbi::ScoreSS (src/hhviterbi.h:193-211) for one aligned column pair, from the index arrays
float score_ss_step(const SsTables& T, float ssw, int mode, const HMMSimd* q_simd, int i, const ResidentTemplate& t,
                    int j) {
  const unsigned qp = q_simd->pred_index[(size_t)(i - 1) * VECSIZE_FLOAT];
  const unsigned qd = q_simd->dssp_index[(size_t)(i - 1) * VECSIZE_FLOAT];
  switch (mode) {
    case HMM::PRED_DSSP:
      return ssw * T.S37[qp / MAXCF][qp % MAXCF][(int)t.ss_dssp[j]];
    case HMM::DSSP_PRED:
      return ssw * T.S73[qd][(int)t.ss_pred[j]][(int)t.ss_conf[j]];
    case HMM::PRED_PRED:
      return ssw * T.S33[qp / MAXCF][qp % MAXCF][(int)t.ss_pred[j]][(int)t.ss_conf[j]];
  }
  return 0.0;
}

}  // namespace

std::vector<Hit> ViterbiRunner::alignment(Parameters& par, HMMSimd* q_simd, std::vector<HHEntry*> dbfiles,
                                          const float qsc, float* pb, const float S[20][20], const float Sim[20][20],
                                          const float R[20][20], const int ssm_mode,
                                          const float S73[NDSSP][NSSPRED][MAXCF],
                                          const float S33[NSSPRED][MAXCF][NSSPRED][MAXCF],
                                          const float S37[NSSPRED][MAXCF][NDSSP]) {
  HMM* q = q_simd->GetHMM(0);
  const int threads = thread_count > 0 ? thread_count : 1;

  // ---- device context = the per-thread Viterbi objects of the reference (src/hhviterbirunner.h:21-34) ----
  hhv_params hp;
  const char* dev = getenv("HHV_DEVICE");
  hp.device = dev ? atoi(dev) : 0;
  hp.local = par.loc;
  hp.egq = par.egq;
  hp.egt = par.egt;
  hp.shift = par.shift;
  hp.corr = par.corr;
  hp.ssw = par.ssw;
  hp.ss_mode = ssm_mode;
  hhv_ctx* ctx = NULL;
  hip_check(hhv_create(&ctx, &hp), "hhv_create");
  {
    Prepared qp;
    lane0_to_profile(q_simd, q, &qp);
    hip_check(hhv_set_query(ctx, qp.p.data(), qp.tr.data(), q->L), "hhv_set_query");
    hip_check(hhv_set_ss_tables(ctx, &S73[0][0][0], &S33[0][0][0][0], &S37[0][0][0]), "hhv_set_ss_tables");
    if (!qp.ss_pred.empty() || !qp.ss_dssp.empty())
      hip_check(hhv_set_query_ss(ctx, qp.ss_pred.empty() ? NULL : qp.ss_pred.data(),
                                 qp.ss_conf.empty() ? NULL : qp.ss_conf.data(),
                                 qp.ss_dssp.empty() ? NULL : qp.ss_dssp.data()),
                "hhv_set_query_ss");
  }
  const SsTables tables = {S73, S33, S37};
  std::vector<int32_t> q_ranges = region_pairs(par.exclstr), t_ranges = region_pairs(par.template_exclstr);
  const bool regions = !q_ranges.empty() || !t_ranges.empty();

  // scratch HMMs, one per thread (the reference keeps VECSIZE_FLOAT per thread, :84-95)
  std::vector<HMM*> t_hmm(threads);
  std::vector<HMMSimd*> t_simd(threads);
  for (int k = 0; k < threads; ++k) {
    t_hmm[k] = new HMM(MAXSEQDIS, par.maxres);
    t_simd[k] = new HMMSimd(par.maxres);
  }

  std::vector<Hit> ret_hits;
  std::vector<hhv_tset*> resident_sets;
  std::map<HHEntry*, ResidentTemplate> resident;
  // earlier alignments per template name: indices into ret_hits (the reference keeps borrowed pointers, :262-268)
  std::map<std::string, std::vector<size_t> > excludeAlignments;
  std::vector<HHEntry*> work(dbfiles.begin(), dbfiles.end());

  // Aligns `members` (positions into `block`, all resident) with one ss mode and appends nothing: fills slot[pos].
  struct Runner {
    hhv_ctx* ctx;
    Parameters& par;
    HMMSimd* q_simd;
    const SsTables& tables;
    const std::vector<int32_t>&q_ranges, &t_ranges;
    bool regions;
    std::map<std::string, std::vector<size_t> >& excl;
    std::vector<Hit>& ret_hits;

    // set: resident chunk; ids: template indices inside it (NULL = the whole set, n templates in set order);
    // tmpl[k]: the resident record of the k-th aligned template; out[k]: its Hit (template information already set)
    void run(hhv_tset* set, const int32_t* ids, int n, int ss_hmm_mode,
             const std::vector<const ResidentTemplate*>& tmpl, const std::vector<Hit*>& out) {
      hhv_tset* ts = set;
      hhv_tset* sub = NULL;
      if (ids) {
        hip_check(hhv_tset_gather(ctx, set, ids, n, &sub), "hhv_tset_gather");
        ts = sub;
      }
      hip_check(hhv_set_ss_mode(ctx, ss_hmm_mode), "hhv_set_ss_mode");
      bool masked = regions;
      if (!excl.empty() || regions) {
        // exclude_alignments (:273-289): every earlier alignment of a template of the same name (also inside round
        // 0: the map is filled block by block, :173)
        std::vector<int32_t> template_of, pi, pj;
        std::vector<int64_t> poff(1, 0);
        for (int k = 0; k < n; ++k) {
          std::map<std::string, std::vector<size_t> >::const_iterator it = excl.find(std::string(out[k]->entry->getName()));
          if (it == excl.end()) continue;
          for (size_t a = 0; a < it->second.size(); ++a) {
            const Hit& h = ret_hits[it->second[a]];
            template_of.push_back(k);
            pi.insert(pi.end(), h.i + 1, h.i + h.nsteps + 1);
            pj.insert(pj.end(), h.j + 1, h.j + h.nsteps + 1);
            poff.push_back((int64_t)pi.size());
            masked = true;
          }
        }
        hip_check(hhv_set_celloff_paths(ctx, ts, (int32_t)template_of.size(), template_of.data(), poff.data(), pi.data(),
                                        pj.data(), (int32_t)q_ranges.size() / 2, q_ranges.data(),
                                        (int32_t)t_ranges.size() / 2, t_ranges.data()),
                  "hhv_set_celloff_paths");
      }
      std::vector<hhv_hit> hits(n);
      hip_check(hhv_align(ctx, ts, masked ? HHV_ALIGN_CELLOFF : HHV_ALIGN_BACKTRACE, NULL), "hhv_align");
      hip_check(hhv_hits(ctx, ts, hits.data()), "hhv_hits");
      for (int k = 0; k < n; ++k) {
        const hhv_hit& h = hits[k];
        Hit& hit = *out[k];
        hit.lastrep = (h.score <= par.smin) ? 1 : 0;  // :37
        hit.realign_around_viterbi = false;
        hit.score = h.score;
        hit.score_ss = h.score_ss;
        hit.score_aass = -h.score;  // BacktraceScore.score_aass, src/hhviterbi.cpp:252
        const int cap = h.nsteps + 1;
        hit.i = new int[cap];
        hit.j = new int[cap];
        hit.states = new char[cap];
        hit.S = new float[cap];
        hit.S_ss = new float[cap];
        int32_t ns = 0;
        hip_check(hhv_hit_path(ctx, ts, k, cap, hit.i, hit.j, (int8_t*)hit.states, hit.S, &ns), "hhv_hit_path");
        hit.i[0] = hit.j[0] = 0;
        hit.states[0] = 0;
        hit.S[0] = hit.S_ss[0] = 0.0f;
        for (int step = 1; step <= h.nsteps; ++step)  // BacktraceScore.S_ss, src/hhviterbi.cpp:222-237
          hit.S_ss[step] = (hit.states[step] == ViterbiMatrix::MM && ss_hmm_mode != HMM::NO_SS_INFORMATION)
                               ? score_ss_step(tables, par.ssw, ss_hmm_mode, q_simd, hit.i[step], *tmpl[k], hit.j[step])
                               : 0.0f;
        hit.nsteps = h.nsteps;
        hit.matched_cols = h.matched_cols;
        hit.i1 = h.i1;
        hit.j1 = h.j1;
        hit.i2 = h.i2;
        hit.j2 = h.j2;
      }
      if (sub) hhv_tset_free(sub);
    }
  } runner = {ctx, par, q_simd, tables, q_ranges, t_ranges, regions, excludeAlignments, ret_hits};

  for (int alignment = 0; alignment < par.altali; alignment++) {
    HH_LOG(INFO) << "Alternative alignment: " << alignment << std::endl;
    const unsigned int n_work = work.size();
    unsigned int block_size = n_work;
    if (alignment == 0 && par.early_stopping_filter) block_size = 2000;  // :109-111
    std::vector<HHEntry*> next_work;

    for (unsigned int block_start = 0; block_start < n_work; block_start += block_size) {
      const unsigned int m = imin(n_work - block_start, block_size);
      sort(work.begin() + block_start, work.begin() + (block_start + m), HHDatabaseEntryCompare());  // :117-119
      const size_t first_hit_of_block = ret_hits.size();
      ret_hits.resize(first_hit_of_block + m);

      // upload in chunks that are whole SIMD batches of the reference, so that host memory stays bounded
      const unsigned int chunk_max = 16384;
      for (unsigned int c0 = 0; c0 < m; c0 += chunk_max) {
        const unsigned int cn = imin(m - c0, chunk_max);
        HHEntry** ent = &work[block_start + c0];
        Hit* hit0 = &ret_hits[first_hit_of_block + c0];

        if (alignment == 0) {
          // ---- read + prepare on the host with the reference's code (:144-147), one template per iteration ----
          std::vector<Prepared> prep(cn);
          std::vector<int> pair_mode(cn);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
          for (unsigned int k = 0; k < cn; ++k) {
            int tid = 0;
#ifdef OPENMP
            tid = omp_get_thread_num();
#endif
            HMM* t = t_hmm[tid];
            int format_tmp = 0;
            char wg = 1;
            ent[k]->getTemplateHMM(par, wg, qsc, format_tmp, pb, S, Sim, t);
            t->entry = ent[k];
            PrepareTemplateHMM(par, q, t, format_tmp, false, pb, R);
            std::vector<HMM*> one(1, t);
            t_simd[tid]->MapHMMVector(one);
            lane0_to_profile(t_simd[tid], t, &prep[k]);
            pair_mode[k] = HMM::computeScoreSSMode(q, t);
            hit0[k].initHitFromHMM(q, t, par.nseqdis, par.ssm);  // :40
            hit0[k].entry = ent[k];
          }
          std::vector<int32_t> L(cn);
          std::vector<const float*> pp(cn), tt(cn);
          std::vector<const int8_t*> sp(cn), sc(cn), sd(cn);
          for (unsigned int k = 0; k < cn; ++k) {
            L[k] = (int32_t)(prep[k].tr.size() / 7) - 1;
            pp[k] = prep[k].p.data();
            tt[k] = prep[k].tr.data();
            sp[k] = prep[k].ss_pred.empty() ? NULL : prep[k].ss_pred.data();
            sc[k] = prep[k].ss_conf.empty() ? NULL : prep[k].ss_conf.data();
            sd[k] = prep[k].ss_dssp.empty() ? NULL : prep[k].ss_dssp.data();
          }
          hhv_tset* set = NULL;
          hip_check(hhv_upload_templates_ss(ctx, (int32_t)cn, L.data(), pp.data(), tt.data(), sp.data(), sc.data(),
                                            sd.data(), &set),
                    "hhv_upload_templates_ss");
          resident_sets.push_back(set);
          for (unsigned int k = 0; k < cn; ++k) {
            ResidentTemplate r;
            r.set = set;
            r.index = (int32_t)k;
            r.L = L[k];
            r.ss_pair_mode = pair_mode[k];
            r.first_hit = first_hit_of_block + c0 + k;
            r.ss_pred.swap(prep[k].ss_pred);
            r.ss_conf.swap(prep[k].ss_conf);
            r.ss_dssp.swap(prep[k].ss_dssp);
            resident[ent[k]] = r;  // an entry listed twice is the same template
          }
        } else {
          for (unsigned int k = 0; k < cn; ++k) {
            copy_template_info(ret_hits[resident[ent[k]].first_hit], &hit0[k]);
            hit0[k].entry = ent[k];
          }
        }

        // ---- the ss mode of every SIMD batch of the reference (:14-22), then one launch per (set, mode) ----
        std::vector<int> batch_mode(cn);
        for (unsigned int b = 0; b < cn; b += VECSIZE_FLOAT) {
          int consensus = 0xFF;
          const unsigned int e = imin(cn, b + VECSIZE_FLOAT);
          for (unsigned int k = b; k < e; ++k) consensus &= resident[ent[k]].ss_pair_mode;
          const int mode = select_ss_mode(consensus);
          for (unsigned int k = b; k < e; ++k) batch_mode[k] = mode;
        }
        std::map<std::pair<hhv_tset*, int>, std::vector<unsigned int> > groups;
        for (unsigned int k = 0; k < cn; ++k)
          groups[std::make_pair(resident[ent[k]].set, batch_mode[k])].push_back(k);
        for (std::map<std::pair<hhv_tset*, int>, std::vector<unsigned int> >::iterator g = groups.begin();
             g != groups.end(); ++g) {
          const std::vector<unsigned int>& mem = g->second;
          const int n = (int)mem.size();
          std::vector<int32_t> ids(n);
          std::vector<const ResidentTemplate*> tmpl(n);
          std::vector<Hit*> out(n);
          bool whole = (n == hhv_tset_size(g->first.first));
          for (int k = 0; k < n; ++k) {
            tmpl[k] = &resident[ent[mem[k]]];
            ids[k] = tmpl[k]->index;
            out[k] = &hit0[mem[k]];
            whole = whole && ids[k] == k;
          }
          runner.run(g->first.first, whole ? NULL : ids.data(), n, g->first.second, tmpl, out);
        }
      }

      // ---- merge_thread_results (:249-271) ----
      for (unsigned int k = 0; k < m; ++k) {
        Hit& h = ret_hits[first_hit_of_block + k];
        h.irep = (alignment + 1);
        if (h.score > par.smin) {
          next_work.push_back(h.entry);
          excludeAlignments[std::string(h.entry->getName())].push_back(first_hit_of_block + k);
        }
      }
      HH_LOG(INFO) << (block_start + m) << " alignments done" << std::endl;

      if (alignment == 0 && par.early_stopping_filter) {  // :178-188
        float early_stopping_sum = calculateEarlyStop(par, q, ret_hits, block_start);
        float filter_cutoff = m * par.filter_thresh;
        if (early_stopping_sum < filter_cutoff) {
          HH_LOG(INFO) << "Stop after DB-HHM: " << (block_start + m) << " because early stop  " << early_stopping_sum
                       << " < filter cutoff " << filter_cutoff << "\n";
          break;
        }
      }
    }
    work.swap(next_work);
  }

  for (size_t k = 0; k < resident_sets.size(); ++k) hhv_tset_free(resident_sets[k]);
  hhv_destroy(ctx);
  for (int k = 0; k < threads; ++k) {
    delete t_simd[k];
    delete t_hmm[k];
  }
  return ret_hits;
}

// src/hhviterbirunner.cpp:213-247.  Sum over the hits of the block of 1 / (1 + E-value), the E-value from the
// neural-network EVD parameters of hhhitlist-inl.h; the expression types (float / double) follow the reference so
// that the comparison with the cutoff cannot fall on the other side.
float ViterbiRunner::calculateEarlyStop(Parameters& par, HMM* q, std::vector<Hit>& all_hits, unsigned int startPos) {
  float sum = 0.0;
  for (unsigned int k = startPos; k < all_hits.size(); k++) {
    const Hit& hit = all_hits[k];
    float q_len = log(q->L) / LOG1000;
    float hit_len = log(hit.L) / LOG1000;
    float q_neff = q->Neff_HMM / 10.0;
    float hit_neff = hit.Neff_HMM / 10.0;
    float lamda = lamda_NN(q_len, hit_len, q_neff, hit_neff);
    float mu = mu_NN(q_len, hit_len, q_neff, hit_neff);
    double logPval = logPvalue(hit.score, lamda, mu);
    float alpha = 0;
    float log_Pcut = log(par.prefilter_evalue_thresh / par.dbsize);
    float log_dbsize = log(par.dbsize);
    if (par.prefilter) alpha = par.alphaa + par.alphab * (hit_neff - 1) * (1 - par.alphac * (q_neff - 1));
    double Eval = exp(logPval + log_dbsize + (alpha * log_Pcut));
    float eval_normalized = 1.0 / (1.0 + Eval);
    sum += eval_normalized;
  }
  return sum;
}
