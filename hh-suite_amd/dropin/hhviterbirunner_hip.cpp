// hhviterbirunner_hip.cpp -- DROP-IN replacement for src/hhviterbirunner.cpp of hh-suite v3.3.0.
//
// A maintainer compiles THIS file instead of src/hhviterbirunner.cpp (same class, same signature, declared by
// the reference's own src/hhviterbirunner.h, which stays untouched) and links libhhviterbi_hip.so.  Every caller
// (HHblits::run x3, RescoreWithViterbiKeepAlignment, HHalign::run) keeps calling
//     ViterbiRunner::alignment(par, q_simd, dbfiles, qsc, pb, S, Sim, R, ssm_mode, S73, S33, S37)
// and receives the same std::vector<Hit> (same fields, same ownership: the path arrays are new[]'d here and freed
// by Hit::Delete).  What moves to the GPU is everything behind Viterbi::Align / Backtrace / ScoreForBacktrace
// (src/hhviterbirunner.cpp:24-31), the exclusion masks (:152-164) and - for templates in HHM format - PrepareTemplateHMM
// (:147), through the C ABI of include/hhviterbi_hip.h.  No DP arithmetic in this file.
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
//
// THE RESIDENT TEMPLATE CACHE.  In the reference more than half of the wall time of this function is
// HHEntry::getTemplateHMM (text parsing) + PrepareTemplateHMM per template per query, and it does not shrink with the
// DP.  Here a template that was read once stays on the device in RAW form (what HMM::Read leaves: frequencies,
// log2 transitions, Neff) for the life of the process, keyed by entry name + sequence length; PrepareTemplateHMM runs
// on the GPU per query (hhv_prepare_subset, bit-identical to the host code: tests/test_prepare.py) because its
// result depends on the query's composition.  From the second time a template is searched - the later iterations of
// hhblits, the next query of hhblits_omp / hhblits_mpi - nothing of it is parsed, prepared or copied by the host;
// its Hit gets the template information (names, displayed sequences) from a prototype kept with the cache entry.
// Templates the device preparation does not cover (HMMER formats, par.pc_hhm_nocontext_mode > 2,
// par.columnscore > 3, a NULL line different from the caller's background) are prepared by the reference's host code
// as before and uploaded prepared; they are not cached.  Environment: HHV_TEMPLATE_CACHE=0 disables the cache,
// HHV_TEMPLATE_CACHE_GB (default 64) bounds it (it is emptied when full), HHV_DEVICE picks the GPU (default 0).
// The cache and its device context are shared by all threads of the process; device work of concurrent callers
// (hhblits_omp) is serialised by a mutex, the host-side reading of new templates is not.
//
// Differences from the reference, by design:
//   * a template is read ONCE per process; the Hit of a later round / later search copies the template information of
//     the prototype instead of calling initHitFromHMM on a freshly read HMM
//   * hits come back in the order of the sorted block (= the reference's order with one thread; with several
//     threads the reference's order depends on the OpenMP schedule)
//   (global mode, par.loc = 0: the reference maximises over the last column of the LONGEST template of each SIMD batch, SURVEY.md 8a
//   row A1; the batches are rebuilt here and the shorter templates marked with hhv_set_global_batch: same results.)
// Errors follow the reference's convention at this layer: HH_LOG(ERROR) + exit(code) (src/hhsearch.h:6).
#include <sys/time.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "hhviterbirunner.h"
#include "hhviterbi_hip.h"
#include "hhv_template_cache.h"
#include "hhv_sidecar.h"

#ifdef OPENMP
#include <omp.h>
#endif

namespace {

using hhv_dropin::SsRecords;
using hhv_dropin::CachedTemplate;
using hhv_dropin::TemplateCache;
using hhv_dropin::cache;
using hhv_dropin::cache_key;

// HHV_DROPIN_TIMING=1 prints where alignment() spends its wall time (stderr)
struct PhaseTimer {
  enum { READ, UPLOAD, PREPARE, MASKS, ALIGN, PATHS, OTHER, N };
  bool on;
  double t_last;
  double acc[N];
  std::vector<std::pair<const char*, double> > detail;  // the same seconds under finer labels (second line of the report)
  size_t cached, fresh, host_prepared, sidecar;
  PhaseTimer() : on(getenv("HHV_DROPIN_TIMING") != NULL), t_last(now()), cached(0), fresh(0), host_prepared(0), sidecar(0) {
    for (int k = 0; k < N; ++k) acc[k] = 0;
  }
  static double now() {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec + 1e-6 * tv.tv_usec;
  }
  void lap(int phase, const char* what = NULL) {
    const double t = now();
    acc[phase] += t - t_last;
    if (on && what) {
      size_t k = 0;
      while (k < detail.size() && detail[k].first != what && strcmp(detail[k].first, what) != 0) ++k;
      if (k == detail.size()) detail.push_back(std::make_pair(what, 0.0));
      detail[k].second += t - t_last;
    }
    t_last = t;
  }
  ~PhaseTimer() {
    if (on)
      fprintf(stderr, "hhviterbirunner_hip: templates %zu cached + %zu read, %zu of them from the sidecar (+ %zu host-prepared); read %.3f s, upload %.3f s, "
              "device prepare %.3f s, masks %.3f s, align+hits %.3f s, paths+Hit %.3f s, other %.3f s\n",
              cached, fresh, sidecar, host_prepared, acc[READ], acc[UPLOAD], acc[PREPARE], acc[MASKS], acc[ALIGN], acc[PATHS], acc[OTHER]);
    if (on && !detail.empty()) {
      fprintf(stderr, "hhviterbirunner_hip:   in ms:");
      for (size_t k = 0; k < detail.size(); ++k) fprintf(stderr, " %s %.2f,", detail[k].first, 1e3 * detail[k].second);
      fprintf(stderr, "\n");
    }
  }
};

void hip_check(int rc, const char* what) {
  if (rc == HHV_OK) return;
  HH_LOG(ERROR) << "hhviterbi_hip: " << what << " failed: " << hhv_last_error() << std::endl;
  exit(rc == HHV_E_MEMORY ? 3 : 4);
}

// host copy of one template on its way to the device: prepared (p, tr) or raw (f, tr, neff)
struct HostTemplate {
  bool raw;
  bool hh_text;  // an .hhm text of a database's hhm ffindex (see CachedTemplate::weights_free)
  int L;
  std::vector<float> p;     // prepared: [(L+1)*20]; raw: f [(L+2)*20]
  std::vector<float> tr;    // [(L+1)*7], enum order of src/hhdecl.h:68
  std::vector<float> neff;  // raw only: [(L+1)*3] Neff_M, Neff_I, Neff_D
  float neff_hmm;
  SsRecords ss;
  int ss_pair_mode;  // HMM::computeScoreSSMode(q, t)
};

// Is the entry of this name an HHM text in every database that has it?  HHblitsDatabase::getEntriesFromNames takes a name
// from hhm_database when it is there (src/hhdatabase.cpp:198-206) and HHEntry::getTemplateHMM recognises the format by
// the first non-blank line (:414-432, "HH..."); alignments (a3m / ca3m entries, or an alignment stored in the hhm index) are
// read through Alignment::FrequenciesAndTransitions, which depends on the weighting argument.
bool entry_is_hh_text(const std::vector<HHblitsDatabase*>& dbs, char* name) {
  bool found = false;
  for (size_t d = 0; d < dbs.size(); ++d) {
    HHblitsDatabase* db = dbs[d];
    if (!db) continue;
    ffindex_entry_t* e = db->hhm_database ? ffindex_get_entry_by_name(db->hhm_database->db_index, name) : NULL;
    if (e) {
      const char* text = ffindex_get_data_by_entry(db->hhm_database->db_data, e);
      size_t at = 0;
      while (text && at < e->length && (text[at] == ' ' || text[at] == '\t' || text[at] == '\n' || text[at] == '\r')) ++at;
      if (!text || at + 2 > e->length || text[at] != 'H' || text[at + 1] != 'H') return false;
      found = true;
      continue;
    }
    if ((db->a3m_database && ffindex_get_entry_by_name(db->a3m_database->db_index, name)) ||
        (db->use_compressed && db->ca3m_database && ffindex_get_entry_by_name(db->ca3m_database->db_index, name)))
      return false;
  }
  return found;
}

// An HHEntry does not say which database it came from (HHDatabaseEntry::ffdatabase / entry are private,
// src/hhdatabase.h:98-101), so everything below finds "the" entry behind a name by looking the name up.  That is only an
// identity while ONE of the searched databases holds the name: with the same name in two -d databases the lookup would
// hand an entry of the second database the text, sidecar record and resident columns of the first.  Such
// names take the reference's own path every time: read through the entry itself, prepared on the host, not cached.
bool name_in_several_databases(const std::vector<HHblitsDatabase*>& dbs, char* name) {
  int holders = 0;
  for (size_t d = 0; d < dbs.size(); ++d) {
    HHblitsDatabase* db = dbs[d];
    if (!db) continue;
    if ((db->hhm_database && ffindex_get_entry_by_name(db->hhm_database->db_index, name)) ||
        (db->a3m_database && ffindex_get_entry_by_name(db->a3m_database->db_index, name)) ||
        (db->use_compressed && db->ca3m_database && ffindex_get_entry_by_name(db->ca3m_database->db_index, name)))
      ++holders;
  }
  return holders > 1;
}

// The hhm ffindex database that holds this entry as an HHM text (and no database holds it as anything else): the
// entries whose parse result the sidecar may stand in for.  *fe = its ffindex entry (offset / length = the validity key).
FFindexDatabase* hh_text_database(const std::vector<HHblitsDatabase*>& dbs, char* name, ffindex_entry_t** fe) {
  if (name_in_several_databases(dbs, name) || !entry_is_hh_text(dbs, name)) return NULL;
  for (size_t d = 0; d < dbs.size(); ++d) {
    HHblitsDatabase* db = dbs[d];
    if (!db || !db->hhm_database) continue;
    ffindex_entry_t* e = ffindex_get_entry_by_name(db->hhm_database->db_index, name);
    if (e) {
      *fe = e;
      return db->hhm_database;
    }
  }
  return NULL;
}

// the database entry behind a name, in the order HHblitsDatabase::getEntriesFromNames looks (src/hhdatabase.cpp:198-215)
// (a name that several databases hold gets an `ambiguous` identity, which equals nothing - not even itself)
hhv_dropin::EntryIdentity identify_entry(const std::vector<HHblitsDatabase*>& dbs, char* name) {
  hhv_dropin::EntryIdentity id;
  if (name_in_several_databases(dbs, name)) {
    id.ambiguous = true;
    return id;
  }
  for (size_t d = 0; d < dbs.size(); ++d) {
    HHblitsDatabase* db = dbs[d];
    if (!db) continue;
    FFindexDatabase* candidates[3] = {db->hhm_database, db->use_compressed ? db->ca3m_database : NULL, db->a3m_database};
    for (int c = 0; c < 3; ++c) {
      if (!candidates[c]) continue;
      ffindex_entry_t* e = ffindex_get_entry_by_name(candidates[c]->db_index, name);
      if (e) {
        id.data = candidates[c]->db_data;
        id.offset = (uint64_t)e->offset;
        id.length = (uint64_t)e->length;
        return id;
      }
    }
  }
  return id;
}

// one Sidecar object per hhm data file of the process
hhv_dropin::Sidecar* sidecar_of(FFindexDatabase* fdb) {
  static std::mutex mu;
  static std::map<std::string, hhv_dropin::Sidecar*> open_files;
  std::lock_guard<std::mutex> lock(mu);
  const std::string path = hhv_dropin::sidecar_path(fdb->data_filename);
  std::map<std::string, hhv_dropin::Sidecar*>::iterator it = open_files.find(path);
  if (it != open_files.end()) return it->second;
  hhv_dropin::Sidecar* sc = new hhv_dropin::Sidecar(path);
  open_files[path] = sc;
  return sc;
}

// what Hit::initHitFromHMM and HMM::computeScoreSSMode read from a template HMM, from a sidecar record into the
// thread's scratch HMM (the sequences of its previous template are released the way HMM::Read does, src/hhhmm.cpp:212-226)
void hmm_header_from_sidecar(const hhv_dropin::SidecarRecord& r, HMM* t) {
  // (dont_delete_seqs, private, is false for these scratch objects: nothing in v3.3.0 sets it, src/hhhit.cpp:205)
  for (int k = 0; k < t->n_seqs; k++) delete[] t->sname[k];
  for (int k = 0; k < t->n_seqs; k++) delete[] t->seq[k];
  t->L = r.L;
  t->Neff_HMM = r.neff_hmm;
  t->n_seqs = r.n_seqs;
  t->n_display = r.n_display;
  t->N_in = t->N_filtered = 0;
  t->ncons = r.ncons;
  t->nfirst = r.nfirst;
  t->nss_dssp = r.nss_dssp;
  t->nsa_dssp = r.nsa_dssp;
  t->nss_pred = r.nss_pred;
  t->nss_conf = r.nss_conf;
  t->lamda = t->mu = 0.0;
  strmcpy(t->name, r.name.c_str(), NAMELEN - 1);
  strmcpy(t->longname, r.longname.c_str(), DESCLEN - 1);
  strmcpy(t->fam, r.fam.c_str(), NAMELEN - 1);
  strmcpy(t->sfam, r.sfam.c_str(), NAMELEN - 1);
  strmcpy(t->fold, r.fold.c_str(), NAMELEN - 1);
  strmcpy(t->cl, r.cl.c_str(), NAMELEN - 1);
  strmcpy(t->file, r.file.c_str(), NAMELEN - 1);
  for (int k = 0; k < r.n_seqs; ++k) {
    t->sname[k] = new char[r.sname[k].size() + 1];
    strcpy(t->sname[k], r.sname[k].c_str());
    t->seq[k] = new char[r.seq[k].size() + 1];
    strcpy(t->seq[k], r.seq[k].c_str());
  }
}

// a template of THIS search: where its prepared columns are on the device
struct ResidentTemplate {
  hhv_tset* set;
  int dev;  // device slot the set lives on
  int32_t index;
  int L;
  int ss_pair_mode;
  size_t first_hit;     // index of its first-round Hit in the result vector (source of the template information)
  const SsRecords* ss;  // in the cache entry or in `own_ss`
  bool raw;             // prepared on the device from raw columns (background pb0), not by the host's PrepareTemplateHMM
};

const int kCacheToEnum[7] = {4 /*I2I*/, 1 /*M2I*/, 0 /*M2M*/, 2 /*M2D*/, 5 /*D2M*/, 6 /*D2D*/, 3 /*I2M*/};

// Lane 0 of an HMMSimd that holds ONE HMM (HMMSimd::MapHMMVector, src/hhhmmsimd.cpp:86-160) -> the profile layout
// of the C ABI: p[(L+1)*20], tr[(L+1)*7] in the enum order of src/hhdecl.h:68.  HMM::tr and the ss arrays are
// private to HMM; HMMSimd is its friend and publishes them, which is how the reference's kernel sees them too.
void lane0_to_profile(const HMMSimd* s, const HMM* h, std::vector<float>* p, std::vector<float>* tr, SsRecords* ss) {
  const int L = h->L;
  p->resize((size_t)(L + 1) * 20);
  tr->resize((size_t)(L + 1) * 7);
  const float* tr_scalar = (const float*)s->tr;
  for (int i = 0; i <= L; ++i) {
    for (int a = 0; a < 20; ++a) (*p)[(size_t)i * 20 + a] = s->p[i][a * VECSIZE_FLOAT];
    for (int c = 0; c < 7; ++c) (*tr)[(size_t)i * 7 + kCacheToEnum[c]] = tr_scalar[((size_t)i * 7 + c) * VECSIZE_FLOAT];
  }
  for (int a = 0; a < 20; ++a) (*p)[a] = 0.0f;  // row 0 is never read by the DP
  ss->pred.clear();
  ss->conf.clear();
  ss->dssp.clear();
  if (h->nss_pred >= 0) {  // pred_index = ss_pred * MAXCF + ss_conf (:133)
    ss->pred.assign(L + 1, 0);
    ss->conf.assign(L + 1, 0);
    for (int i = 1; i <= L; ++i) {
      const unsigned v = s->pred_index[(size_t)(i - 1) * VECSIZE_FLOAT];
      ss->pred[i] = (int8_t)(v / MAXCF);
      ss->conf[i] = (int8_t)(v % MAXCF);
    }
  }
  if (h->nss_dssp >= 0) {
    ss->dssp.assign(L + 1, 0);
    for (int i = 1; i <= L; ++i) ss->dssp[i] = (int8_t)s->dssp_index[(size_t)(i - 1) * VECSIZE_FLOAT];
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

// A Hit of a template whose information another Hit already carries: every owned array is duplicated
// (Hit::Delete frees them per Hit, src/hhhit.cpp:38-62)
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
float score_ss_step(const SsTables& T, float ssw, int mode, const HMMSimd* q_simd, int i, const SsRecords& t, int j) {
  const unsigned qp = q_simd->pred_index[(size_t)(i - 1) * VECSIZE_FLOAT];
  const unsigned qd = q_simd->dssp_index[(size_t)(i - 1) * VECSIZE_FLOAT];
  switch (mode) {
    case HMM::PRED_DSSP:
      return ssw * T.S37[qp / MAXCF][qp % MAXCF][(int)t.dssp[j]];
    case HMM::DSSP_PRED:
      return ssw * T.S73[qd][(int)t.pred[j]][(int)t.conf[j]];
    case HMM::PRED_PRED:
      return ssw * T.S33[qp / MAXCF][qp % MAXCF][(int)t.pred[j]][(int)t.conf[j]];
  }
  return 0.0;
}

// One search = one object: the caller's arguments, the query as the device wants it, and the device sections.
struct Search {
  Parameters& par;
  HMMSimd* q_simd;
  HMM* q;
  const int ssm_mode;
  const SsTables tables;
  TemplateCache& tc;
  unsigned long id;
  int threads;
  PhaseTimer timer;
  std::vector<float> q_p, q_tr;
  SsRecords q_ss;
  std::vector<int32_t> q_ranges, t_ranges;
  bool regions;
  std::vector<Hit> ret_hits;
  std::map<std::string, std::vector<size_t> > excludeAlignments;  // earlier alignments per template name (:262-268)

  Search(Parameters& par, HMMSimd* q_simd, int ssm_mode, const SsTables& tables)
      : par(par), q_simd(q_simd), q(q_simd->GetHMM(0)), ssm_mode(ssm_mode), tables(tables), tc(cache()), id(0), threads(1), regions(false) {
    lane0_to_profile(q_simd, q, &q_p, &q_tr, &q_ss);
    q_ranges = region_pairs(par.exclstr);
    t_ranges = region_pairs(par.template_exclstr);
    regions = !q_ranges.empty() || !t_ranges.empty();
  }

  // Enter a device section (tc.device is held by the caller): every device's context gets this search's parameters and
  // query unless it still has them from the previous section.
  void install() {
    hip_check(hhv_dropin::ensure_context(tc), "hhv_create");
    if (id == 0) id = ++tc.calls;
    for (size_t d = 0; d < tc.slots.size(); ++d) {
      hhv_dropin::DeviceSlot& sl = tc.slots[d];
      if (sl.owner == id) continue;
      hhv_params hp;
      hp.device = sl.device_id;
      hp.local = par.loc;
      hp.egq = par.egq;
      hp.egt = par.egt;
      hp.shift = par.shift;
      hp.corr = par.corr;
      hp.ssw = par.ssw;
      hp.ss_mode = ssm_mode;  // the ss_mode argument of the Viterbi constructor (src/hhviterbirunner.h:32-33)
      hip_check(hhv_set_params(sl.ctx, &hp), "hhv_set_params");
      hip_check(hhv_set_query(sl.ctx, q_p.data(), q_tr.data(), q->L), "hhv_set_query");
      hip_check(hhv_set_ss_tables(sl.ctx, &tables.S73[0][0][0], &tables.S33[0][0][0][0], &tables.S37[0][0][0]), "hhv_set_ss_tables");
      hip_check(hhv_set_query_ss(sl.ctx, q_ss.pred.empty() ? NULL : q_ss.pred.data(), q_ss.conf.empty() ? NULL : q_ss.conf.data(),
                                 q_ss.dssp.empty() ? NULL : q_ss.dssp.data()),
                "hhv_set_query_ss");
      sl.owner = id;
    }
  }

  // Aligns n templates of one resident set with one ss mode (device section).  ids: their indices in the set (NULL =
  // the whole set in set order); tmpl[k] / out[k]: resident record and Hit (template information already set) of the k-th.
  // dev: the device slot the set lives on; timed: this call runs on the search's own thread (the phase timer is not shared
  // between the per-device threads of a multi-device search)
  void run(int dev, bool timed, hhv_tset* set, const int32_t* ids, int n, int ss_hmm_mode,
           const std::vector<const ResidentTemplate*>& tmpl, const std::vector<Hit*>& out, const std::vector<uint8_t>& not_longest) {
    hhv_ctx* ctx = tc.slots[dev].ctx;
    hhv_tset* ts = set;
    hhv_tset* sub = NULL;
    if (timed) timer.lap(PhaseTimer::OTHER, "groups");
    if (ids) {
      hip_check(hhv_tset_gather(ctx, set, ids, n, &sub), "hhv_tset_gather");
      ts = sub;
    }
    hip_check(hhv_set_ss_mode(ctx, ss_hmm_mode), "hhv_set_ss_mode");
    // global mode: the shorter templates of a SIMD batch of the reference do not see their own last column (:462-486)
    if (!par.loc) hip_check(hhv_set_global_batch(ctx, ts, not_longest.data()), "hhv_set_global_batch");
    bool masked = regions;
    if (!excludeAlignments.empty() || regions) {
      // exclude_alignments (:273-289): every earlier alignment of a template of the same name (also inside round
      // 0: the map is filled block by block, :173)
      std::vector<int32_t> template_of, pi, pj;
      std::vector<int64_t> poff(1, 0);
      for (int k = 0; k < n && !excludeAlignments.empty(); ++k) {
        std::map<std::string, std::vector<size_t> >::const_iterator it = excludeAlignments.find(std::string(out[k]->entry->getName()));
        if (it == excludeAlignments.end()) continue;
        for (size_t a = 0; a < it->second.size(); ++a) {
          const Hit& h = ret_hits[it->second[a]];
          template_of.push_back(k);
          pi.insert(pi.end(), h.i + 1, h.i + h.nsteps + 1);
          pj.insert(pj.end(), h.j + 1, h.j + h.nsteps + 1);
          poff.push_back((int64_t)pi.size());
          masked = true;
        }
      }
      hip_check(hhv_set_celloff_paths(ctx, ts, (int32_t)template_of.size(), template_of.data(), poff.data(), pi.data(), pj.data(),
                                      (int32_t)q_ranges.size() / 2, q_ranges.data(), (int32_t)t_ranges.size() / 2,
                                      t_ranges.data()),
                "hhv_set_celloff_paths");
    }
    std::vector<hhv_hit> hits(n);
    if (timed) timer.lap(PhaseTimer::MASKS, "gather+masks");
    hip_check(hhv_align(ctx, ts, masked ? HHV_ALIGN_CELLOFF : HHV_ALIGN_BACKTRACE, NULL), "hhv_align");
    if (timed) timer.lap(PhaseTimer::ALIGN, "hhv_align");
    hip_check(hhv_hits(ctx, ts, hits.data()), "hhv_hits");
    if (timed) timer.lap(PhaseTimer::ALIGN, "hhv_hits");
    // the paths in one piece: the compact records of hhv_hit_paths_packed (steps 0 .. nsteps of every hit, 16-bit i / j, in a
    // pinned buffer of the context), else the host mirror of the whole pool, else hit by hit; the Hit objects are filled by all
    // threads
    const int64_t *pk_off = NULL, *path_off = NULL;
    const uint16_t *pk_i = NULL, *pk_j = NULL;
    const int32_t *pool_i = NULL, *pool_j = NULL;
    const int8_t *pk_states = NULL, *pool_states = NULL;
    const float *pk_S = NULL, *pool_S = NULL;
    const bool packed = hhv_hit_paths_packed(ctx, ts, hits.data(), &pk_off, &pk_i, &pk_j, &pk_states, &pk_S) == HHV_OK;
    const bool pooled = !packed && hhv_hit_path_pool(ctx, ts, &path_off, &pool_i, &pool_j, &pool_states, &pool_S) == HHV_OK;
    const bool with_ss = ss_hmm_mode != HMM::NO_SS_INFORMATION;
    if (timed) timer.lap(PhaseTimer::PATHS, "paths to host");
#pragma omp parallel for schedule(static) num_threads(threads) if ((packed || pooled) && n > 256)
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
      if (packed) {
        const int64_t po = pk_off[k];
        const uint16_t *si = pk_i + po, *sj = pk_j + po;
        for (int s = 0; s < cap; ++s) {
          hit.i[s] = si[s];
          hit.j[s] = sj[s];
        }
        memcpy(hit.states, pk_states + po, (size_t)cap);
        memcpy(hit.S, pk_S + po, (size_t)cap * sizeof(float));
      } else if (pooled) {
        const int64_t po = path_off[k];
        memcpy(hit.i, pool_i + po, (size_t)cap * sizeof(int));
        memcpy(hit.j, pool_j + po, (size_t)cap * sizeof(int));
        memcpy(hit.states, pool_states + po, (size_t)cap);
        memcpy(hit.S, pool_S + po, (size_t)cap * sizeof(float));
      } else {
        int32_t ns = 0;
        hip_check(hhv_hit_path(ctx, ts, k, cap, hit.i, hit.j, (int8_t*)hit.states, hit.S, &ns), "hhv_hit_path");
      }
      hit.i[0] = hit.j[0] = 0;
      hit.states[0] = 0;
      hit.S[0] = hit.S_ss[0] = 0.0f;
      if (with_ss) {
        for (int step = 1; step <= h.nsteps; ++step)  // BacktraceScore.S_ss, src/hhviterbi.cpp:222-237
          hit.S_ss[step] = hit.states[step] == ViterbiMatrix::MM
                               ? score_ss_step(tables, par.ssw, ss_hmm_mode, q_simd, hit.i[step], *tmpl[k]->ss, hit.j[step])
                               : 0.0f;
      } else {
        memset(hit.S_ss, 0, (size_t)cap * sizeof(float));
      }
      hit.nsteps = h.nsteps;
      hit.matched_cols = h.matched_cols;
      hit.i1 = h.i1;
      hit.j1 = h.j1;
      hit.i2 = h.i2;
      hit.j2 = h.j2;
    }
    if (timed) timer.lap(PhaseTimer::PATHS, "Hit arrays");
    if (sub) hhv_tset_free(sub);
    if (timed) timer.lap(PhaseTimer::PATHS, "free subset");
  }
};

}  // namespace

// Housekeeping for the embedding program (optional): empty the resident template cache (e.g. after the database files
// were rebuilt) and look at its size.  Safe to call between searches; a running search keeps the cache alive.
extern "C" void hhviterbirunner_hip_cache_clear() {
  TemplateCache& tc = cache();
  std::lock_guard<std::mutex> lock(tc.device);
  if (tc.active == 0) tc.clear();
}
extern "C" void hhviterbirunner_hip_cache_stats(size_t* templates, size_t* columns) {
  TemplateCache& tc = cache();
  std::lock_guard<std::mutex> lock(tc.device);
  if (templates) *templates = tc.map.size();
  if (columns) *columns = tc.columns;
}

std::vector<Hit> ViterbiRunner::alignment(Parameters& par, HMMSimd* q_simd, std::vector<HHEntry*> dbfiles,
                                          const float qsc, float* pb, const float S[20][20], const float Sim[20][20],
                                          const float R[20][20], const int ssm_mode,
                                          const float S73[NDSSP][NSSPRED][MAXCF],
                                          const float S33[NSSPRED][MAXCF][NSSPRED][MAXCF],
                                          const float S37[NSSPRED][MAXCF][NDSSP]) {
  const SsTables tables = {S73, S33, S37};
  Search search(par, q_simd, ssm_mode, tables);
  HMM* q = search.q;
  TemplateCache& tc = search.tc;
  PhaseTimer& timer = search.timer;
  std::vector<Hit>& ret_hits = search.ret_hits;
  const int threads = thread_count > 0 ? thread_count : 1;
  search.threads = threads;

  bool device_prepare = tc.enabled && hhv_dropin::device_prepare_covers(par);
  const bool use_sidecar = device_prepare && hhv_dropin::sidecar_enabled();
  // The background raw (device-prepared) templates are prepared against.  HMM::Read overwrites pb with the NULL line of every file it
  // reads, ReadHMMer / ReadHMMer3 with the file's NULE / COMPO line, and PrepareTemplateHMM follows the read: every template meets
  // its own file's background.  A template is taken raw when that background is the cache's (hhv_template_cache.h null_pb; the
  // caller's pb when the cache is empty) - the others are prepared by the host with theirs.  (Until the last session of round 6
  // this was the caller's pb at entry: after a search that had read an HMMER-format template last, cached templates were prepared
  // against that file's COMPO line - tests/test_dropin_apps.py::test_hhsearch_database_with_hmmer3_templates.)
  float pb0[20];
  memcpy(pb0, pb, sizeof(pb0));

  // scratch HMMs, one per thread (the reference keeps VECSIZE_FLOAT per thread, :84-95), allocated when a template
  // has to be read
  std::vector<HMM*> t_hmm(threads, (HMM*)NULL);
  std::vector<HMMSimd*> t_simd(threads, (HMMSimd*)NULL);
  std::vector<HMM*> t_hdr(threads, (HMM*)NULL);  // header-only HMMs of the templates that come from the sidecar

  std::vector<hhv_tset*> search_sets;              // prepared sets of this search, freed at the end
  std::unordered_map<HHEntry*, ResidentTemplate> resident;   // an entry listed twice is the same template
  std::vector<SsRecords*> own_ss;                  // ss records of host-prepared templates
  std::vector<HHEntry*> work(dbfiles.begin(), dbfiles.end());
  resident.reserve(dbfiles.size());

  if (device_prepare) {  // the prototypes of the cache are valid for one (nseqdis, ssm, query ss presence) only
    std::lock_guard<std::mutex> lock(tc.device);
    const int qp = q->nss_pred >= 0, qd = q->nss_dssp >= 0;
    const uint64_t rph = hhv_dropin::read_parameters_hash(par, qsc);
    if (tc.nseqdis != par.nseqdis || tc.ssm != par.ssm || tc.q_has_pred != qp || tc.q_has_dssp != qd || tc.read_param_hash != rph) {
      if (tc.active == 0) {
        tc.clear();
        tc.nseqdis = par.nseqdis;
        tc.ssm = par.ssm;
        tc.q_has_pred = qp;
        tc.q_has_dssp = qd;
        tc.read_param_hash = rph;
      } else {
        device_prepare = false;  // a concurrent search of another kind is using the cache: this one prepares on the host
      }
    }
    if (device_prepare) {
      tc.active++;
      if (tc.null_pb_set) memcpy(pb0, tc.null_pb, sizeof(pb0));
    }
  }
  hhv_prep_params prep = hhv_dropin::prepare_params(par, pb0, R);

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
      timer.lap(PhaseTimer::OTHER, "sort");
      // (a Hit is ~1.4 KB - four IDLEN strings: 20 000 of them are 28 MB, 2 ms of constructors on warm pages and 6-12 ms when the
      // allocator has to map fresh ones; touching the pages from all threads first was measured and is slower)
      ret_hits.resize(first_hit_of_block + m);

      // chunks that are whole SIMD batches of the reference, so that host memory stays bounded
      // (32 768: hhblits' 20 000 templates of a round in one piece - every chunk is a round of launches, copies and waits)
      const unsigned int chunk_max = 32768;
      for (unsigned int c0 = 0; c0 < m; c0 += chunk_max) {
        const unsigned int cn = imin(m - c0, chunk_max);
        HHEntry** ent = &work[block_start + c0];
        Hit* hit0 = &ret_hits[first_hit_of_block + c0];
        timer.lap(PhaseTimer::OTHER, "Hit vector");

        if (alignment == 0) {
          // ---- which templates are already resident in raw form? ----
          std::vector<const CachedTemplate*> cached(cn, (const CachedTemplate*)NULL);
          std::vector<unsigned int> to_read;
          if (device_prepare) {
            std::lock_guard<std::mutex> lock(tc.device);
            // (lookups only - nobody writes the map while the lock is held - by all threads: key string, hash and the index
            // search of the entry's name are ~0.25 us a template)
#pragma omp parallel for schedule(static) num_threads(threads) if (cn > 256)
            for (unsigned int k = 0; k < cn; ++k) {
              std::unordered_map<std::string, CachedTemplate>::const_iterator it = tc.map.find(cache_key(ent[k]));
              // same name and length but another database entry (two databases, a rebuilt one): read it again, the new
              // upload takes the slot
              const hhv_dropin::EntryIdentity id = identify_entry(databases, ent[k]->getName());
              if (it != tc.map.end() && it->second.id == id) cached[k] = &it->second;  // std::unordered_map never moves its elements
            }
            for (unsigned int k = 0; k < cn; ++k)
              if (!cached[k]) to_read.push_back(k);
          } else {
            for (unsigned int k = 0; k < cn; ++k) to_read.push_back(k);
          }
          timer.cached += cn - to_read.size();
          timer.lap(PhaseTimer::READ, "cache lookup");

          // ---- read the others with the reference's code (:144), one template per iteration; no device lock ----
          std::vector<HostTemplate> host(to_read.size());
          const int n_read = (int)to_read.size();
          std::vector<char> from_sidecar(to_read.size(), 0);
          std::vector<hhv_dropin::Sidecar*> touched(to_read.size(), (hhv_dropin::Sidecar*)NULL);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
          for (int r = 0; r < n_read; ++r) {
            int tid = 0;
#ifdef OPENMP
            tid = omp_get_thread_num();
#endif
            const unsigned int k = to_read[r];
            HostTemplate& h = host[r];
            int format_tmp = 0;
            char wg = 1;
            // the binary sidecar of the database first (hhv_sidecar.h): an HHM text that was parsed once - by any process -
            // is not parsed again
            ffindex_entry_t* fe = NULL;
            FFindexDatabase* fdb = use_sidecar ? hh_text_database(databases, ent[k]->getName(), &fe) : NULL;
            hhv_dropin::Sidecar* sc = fdb ? sidecar_of(fdb) : NULL;
            hhv_dropin::SidecarRecord rec;
            const uint64_t text_hash = sc ? hhv_dropin::sidecar_text_hash(ffindex_get_data_by_entry(fdb->db_data, fe), fe->length) : 0;
            if (sc && sc->find(ent[k]->getName(), text_hash, (uint64_t)fe->length, par.nseqdis, pb0, &rec) &&
                rec.L + 2 <= par.maxres && rec.n_seqs <= MAXSEQDIS) {
              // a header-only HMM: the full-size scratch HMM (par.maxres rows, ~80 000 allocations) is only built by a
              // thread that really has to parse
              if (!t_hdr[tid]) t_hdr[tid] = new HMM(MAXSEQDIS, 2);
              HMM* t = t_hdr[tid];
              hmm_header_from_sidecar(rec, t);
              t->entry = ent[k];
              h.raw = h.hh_text = true;
              h.L = rec.L;
              h.ss_pair_mode = HMM::computeScoreSSMode(q, t);
              hit0[k].initHitFromHMM(q, t, par.nseqdis, par.ssm);  // :40
              h.p.swap(rec.f);
              h.tr.swap(rec.tr);
              h.neff.swap(rec.neff);
              h.neff_hmm = rec.neff_hmm;
              h.ss.pred.swap(rec.pred);
              h.ss.conf.swap(rec.conf);
              h.ss.dssp.swap(rec.dssp);
              from_sidecar[r] = 1;
              continue;
            }
            if (!t_hmm[tid]) {
              t_hmm[tid] = new HMM(MAXSEQDIS, par.maxres);
              t_simd[tid] = new HMMSimd(par.maxres);
            }
            HMM* t = t_hmm[tid];
            ent[k]->getTemplateHMM(par, wg, qsc, format_tmp, pb, S, Sim, t);
            t->entry = ent[k];
            h.raw = device_prepare && format_tmp == 0 && memcmp(pb, pb0, sizeof(pb0)) == 0 && t->L >= 1 && t->L <= 0xFFFF &&
                    !name_in_several_databases(databases, ent[k]->getName());
            h.L = t->L;
            h.hh_text = h.raw && entry_is_hh_text(databases, ent[k]->getName());
            h.ss_pair_mode = HMM::computeScoreSSMode(q, t);
            hit0[k].initHitFromHMM(q, t, par.nseqdis, par.ssm);  // :40
            std::vector<HMM*> one(1, t);
            if (h.raw) {
              // raw columns as HMM::Read leaves them: p := f (HMM::NoAminoAcidPseudocounts), transitions untouched
              t->NoAminoAcidPseudocounts();
              t_simd[tid]->MapHMMVector(one);
              std::vector<float> f;
              lane0_to_profile(t_simd[tid], t, &f, &h.tr, &h.ss);
              h.p.assign((size_t)(t->L + 2) * 20, 0.0f);
              memcpy(h.p.data() + 20, f.data() + 20, (size_t)t->L * 20 * sizeof(float));
              h.neff.resize((size_t)(t->L + 1) * 3);
              for (int i = 0; i <= t->L; ++i) {
                h.neff[(size_t)i * 3 + 0] = t->Neff_M[i];
                h.neff[(size_t)i * 3 + 1] = t->Neff_I[i];
                h.neff[(size_t)i * 3 + 2] = t->Neff_D[i];
              }
              h.neff_hmm = t->Neff_HMM;
              // parsed here for the first time: leave it in the sidecar for the next process - unless this parse may have been
              // cut short by the limits of THIS run (HMM::Read stops at maxres - 2 columns and at maxcol - 1 characters per
              // sequence, src/hhhmm.cpp:395-468,601,669): a record is valid for every later run, whatever its -maxres
              bool whole = t->L < par.maxres - 2;
              for (int x = 0; whole && x < t->n_seqs; ++x) whole = t->seq[x] == NULL || (int)strlen(t->seq[x]) < par.maxcol - 2;
              if (sc && h.hh_text && whole) {
                rec.ff_hash = text_hash;
                rec.ff_length = (uint64_t)fe->length;
                rec.nseqdis = par.nseqdis;
                memcpy(rec.null_pb, pb0, sizeof(rec.null_pb));
                rec.L = t->L;
                rec.n_seqs = t->n_seqs;
                rec.n_display = t->n_display;
                rec.ncons = t->ncons;
                rec.nfirst = t->nfirst;
                rec.nss_dssp = t->nss_dssp;
                rec.nsa_dssp = t->nsa_dssp;
                rec.nss_pred = t->nss_pred;
                rec.nss_conf = t->nss_conf;
                rec.neff_hmm = t->Neff_HMM;
                rec.entry_name = ent[k]->getName();
                rec.name = t->name;
                rec.longname = t->longname;
                rec.fam = t->fam;
                rec.sfam = t->sfam;
                rec.fold = t->fold;
                rec.cl = t->cl;
                rec.file = t->file;
                rec.sname.resize(t->n_seqs);
                rec.seq.resize(t->n_seqs);
                for (int x = 0; x < t->n_seqs; ++x) {
                  rec.sname[x] = t->sname[x];
                  rec.seq[x] = t->seq[x];
                }
                rec.f = h.p;
                rec.tr = h.tr;
                rec.neff = h.neff;
                rec.pred = h.ss.pred;
                rec.conf = h.ss.conf;
                rec.dssp = h.ss.dssp;
                sc->add(rec);
                touched[r] = sc;
              }
            } else {
              PrepareTemplateHMM(par, q, t, format_tmp, false, pb, R);  // :147
              t_simd[tid]->MapHMMVector(one);
              lane0_to_profile(t_simd[tid], t, &h.p, &h.tr, &h.ss);
            }
          }
          {  // what was parsed for the first time goes to the sidecar files: one append per file and search
            std::vector<hhv_dropin::Sidecar*> files;
            for (int r = 0; r < n_read; ++r) {
              timer.sidecar += from_sidecar[r];
              if (touched[r] && std::find(files.begin(), files.end(), touched[r]) == files.end()) files.push_back(touched[r]);
            }
            for (size_t x = 0; x < files.size(); ++x) files[x]->flush();
          }
          timer.lap(PhaseTimer::READ, "read+parse");

          // ---- device section 1: new raw templates into the cache, host-prepared ones into a set of this search,
          //      PrepareTemplateHMM on the device for everything raw ----
          {
            std::lock_guard<std::mutex> lock(tc.device);
            search.install();
            const int n_slots = (int)tc.slots.size();
            hhv_ctx* ctx = tc.slots[0].ctx;  // host-prepared templates (the rare path) go to the primary device
            std::vector<unsigned int> raw_k, prep_k;
            std::vector<int> raw_r, prep_r;
            for (int r = 0; r < n_read; ++r) {
              if (host[r].raw) {
                raw_k.push_back(to_read[r]);
                raw_r.push_back(r);
              } else {
                prep_k.push_back(to_read[r]);
                prep_r.push_back(r);
              }
            }
            timer.fresh += raw_k.size();
            timer.host_prepared += prep_k.size();
            if (!raw_k.empty()) {
              size_t cols = 0;
              const int n = (int)raw_k.size();
              std::vector<int32_t> L(n);
              for (int x = 0; x < n; ++x) {
                L[x] = host[raw_r[x]].L;
                cols += (size_t)L[x] + 1;
              }
              // A full cache is emptied only when no template of it is in use by this search (nothing resident yet,
              // nothing of this chunk found in it); otherwise it grows past the bound until the search is over.
              if (tc.columns + cols > tc.max_columns && tc.active == 1 && resident.empty() && to_read.size() == cn) tc.clear();
              // the new templates are spread over the devices (hhv_shard_plan: length-sorted bins, longest processing time
              // first); a template stays on the device that first received it for as long as it is cached
              std::vector<int32_t> shard_of(n, 0);
              if (n_slots > 1) hip_check(hhv_shard_plan(n, L.data(), n_slots, shard_of.data()), "hhv_shard_plan");
              std::vector<std::vector<int> > members(n_slots);
              for (int x = 0; x < n; ++x) members[shard_of[x]].push_back(x);
              std::vector<hhv_rawset*> new_set(n_slots, (hhv_rawset*)NULL);
              std::vector<std::thread> uploads;
              for (int d = 0; d < n_slots; ++d) {
                if (members[d].empty()) continue;
                uploads.push_back(std::thread([&, d]() {  // one host thread per context
                  const std::vector<int>& mem = members[d];
                  const int m = (int)mem.size();
                  std::vector<int32_t> Ld(m);
                  std::vector<const float*> ff(m), tt(m), ne(m);
                  std::vector<float> nh(m);
                  std::vector<const int8_t*> sp(m), sc(m), sd(m);
                  for (int y = 0; y < m; ++y) {
                    const HostTemplate& h = host[raw_r[mem[y]]];
                    Ld[y] = h.L;
                    ff[y] = h.p.data();
                    tt[y] = h.tr.data();
                    ne[y] = h.neff.data();
                    nh[y] = h.neff_hmm;
                    sp[y] = h.ss.pred.empty() ? NULL : h.ss.pred.data();
                    sc[y] = h.ss.conf.empty() ? NULL : h.ss.conf.data();
                    sd[y] = h.ss.dssp.empty() ? NULL : h.ss.dssp.data();
                  }
                  hip_check(hhv_upload_raw_templates(tc.slots[d].ctx, m, Ld.data(), ff.data(), tt.data(), ne.data(), nh.data(), sp.data(),
                                                     sc.data(), sd.data(), &new_set[d]),
                            "hhv_upload_raw_templates");
                }));
              }
              for (size_t u = 0; u < uploads.size(); ++u) uploads[u].join();
              tc.columns += cols;
              if (!tc.null_pb_set) {  // (every raw template of this search was read with pb0, see h.raw)
                memcpy(tc.null_pb, pb0, sizeof(tc.null_pb));
                tc.null_pb_set = true;
              }
              std::vector<CachedTemplate*> slot(n);
              // the same key twice in one chunk (an entry listed twice): both x share ONE CachedTemplate; the later upload
              // owns it and only that x fills it below (two fills would race on ct.proto under the parallel loop)
              std::vector<char> fills(n, 0);
              std::unordered_map<CachedTemplate*, int> owner;
              for (int d = 0; d < n_slots; ++d) {
                if (members[d].empty()) continue;
                tc.slots[d].rawsets.push_back(new_set[d]);
                for (size_t y = 0; y < members[d].size(); ++y) {  // std::unordered_map never moves its elements
                  const int x = members[d][y];
                  tc.slots[d].columns += (size_t)L[x] + 1;
                  CachedTemplate& ct = tc.map[cache_key(ent[raw_k[x]])];
                  if (ct.raw) ct.proto.Delete();  // the same key twice (two searches read it concurrently): the later upload wins
                  ct.raw = new_set[d];
                  ct.dev = d;
                  ct.index = (int32_t)y;
                  ct.id = identify_entry(databases, ent[raw_k[x]]->getName());
                  slot[x] = &ct;
                  cached[raw_k[x]] = &ct;
                  std::unordered_map<CachedTemplate*, int>::iterator ow = owner.find(&ct);
                  if (ow != owner.end()) fills[ow->second] = 0;
                  owner[&ct] = x;
                  fills[x] = 1;
                }
              }
#pragma omp parallel for schedule(static) num_threads(threads) if (n > 256)
              for (int x = 0; x < n; ++x) {
                HostTemplate& h = host[raw_r[x]];
                CachedTemplate& ct = *slot[x];
                if (!fills[x]) continue;
                ct.L = h.L;
                ct.weights_free = h.hh_text;
                ct.ss_pair_mode = h.ss_pair_mode;
                copy_template_info(hit0[raw_k[x]], &ct.proto);
                ct.proto.entry = NULL;
                ct.ss.pred.swap(h.ss.pred);
                ct.ss.conf.swap(h.ss.conf);
                ct.ss.dssp.swap(h.ss.dssp);
              }
            }
            timer.lap(PhaseTimer::UPLOAD);
            if (!prep_k.empty()) {
              const int n = (int)prep_k.size();
              std::vector<int32_t> L(n);
              std::vector<const float*> pp(n), tt(n);
              std::vector<const int8_t*> sp(n), sc(n), sd(n);
              for (int x = 0; x < n; ++x) {
                HostTemplate& h = host[prep_r[x]];
                L[x] = h.L;
                pp[x] = h.p.data();
                tt[x] = h.tr.data();
                sp[x] = h.ss.pred.empty() ? NULL : h.ss.pred.data();
                sc[x] = h.ss.conf.empty() ? NULL : h.ss.conf.data();
                sd[x] = h.ss.dssp.empty() ? NULL : h.ss.dssp.data();
              }
              hhv_tset* set = NULL;
              hip_check(hhv_upload_templates_ss(ctx, n, L.data(), pp.data(), tt.data(), sp.data(), sc.data(), sd.data(), &set),
                        "hhv_upload_templates_ss");
              search_sets.push_back(set);
              for (int x = 0; x < n; ++x) {
                HostTemplate& h = host[prep_r[x]];
                SsRecords* ss = new SsRecords();
                ss->pred.swap(h.ss.pred);
                ss->conf.swap(h.ss.conf);
                ss->dssp.swap(h.ss.dssp);
                own_ss.push_back(ss);
                ResidentTemplate rt = {set, 0, (int32_t)x, h.L, h.ss_pair_mode, first_hit_of_block + c0 + prep_k[x], ss, false};
                resident[ent[prep_k[x]]] = rt;
              }
            }
            timer.lap(PhaseTimer::UPLOAD);
            // PrepareTemplateHMM on the device, one launch per raw set the chunk's templates live in; the devices work side
            // by side (one host thread each)
            std::map<hhv_rawset*, std::vector<unsigned int> > by_raw;
            for (unsigned int k = 0; k < cn; ++k)
              if (cached[k]) by_raw[cached[k]->raw].push_back(k);
            std::vector<std::vector<hhv_rawset*> > raw_of_slot(n_slots);
            for (std::map<hhv_rawset*, std::vector<unsigned int> >::iterator g = by_raw.begin(); g != by_raw.end(); ++g)
              raw_of_slot[cached[g->second[0]]->dev].push_back(g->first);
            std::map<hhv_rawset*, hhv_tset*> prepared;
            for (std::map<hhv_rawset*, std::vector<unsigned int> >::iterator g = by_raw.begin(); g != by_raw.end(); ++g) prepared[g->first] = NULL;
            auto prepare_slot = [&](int d) {
              for (size_t w = 0; w < raw_of_slot[d].size(); ++w) {
                hhv_rawset* rs = raw_of_slot[d][w];
                const std::vector<unsigned int>& mem = by_raw[rs];
                std::vector<int32_t> ids(mem.size());
                for (size_t x = 0; x < mem.size(); ++x) ids[x] = cached[mem[x]]->index;
                hip_check(hhv_prepare_subset(tc.slots[d].ctx, rs, &prep, q->pav, ids.data(), (int32_t)ids.size(), &prepared[rs]),
                          "hhv_prepare_subset");
              }
            };
            if (n_slots == 1) {
              prepare_slot(0);
            } else {
              std::vector<std::thread> workers;
              for (int d = 0; d < n_slots; ++d)
                if (!raw_of_slot[d].empty()) workers.push_back(std::thread(prepare_slot, d));
              for (size_t w = 0; w < workers.size(); ++w) workers[w].join();
            }
            for (std::map<hhv_rawset*, std::vector<unsigned int> >::iterator g = by_raw.begin(); g != by_raw.end(); ++g) {
              const std::vector<unsigned int>& mem = g->second;
              hhv_tset* set = prepared[g->first];
              search_sets.push_back(set);
              for (size_t x = 0; x < mem.size(); ++x) {
                const unsigned int k = mem[x];
                const CachedTemplate* ct = cached[k];
                ResidentTemplate rt = {set, ct->dev, (int32_t)x, ct->L, ct->ss_pair_mode, first_hit_of_block + c0 + k, &ct->ss, true};
                resident[ent[k]] = rt;
              }
            }
            timer.lap(PhaseTimer::PREPARE, "device prepare");
            // templates not read in this search: template information from the prototype
#pragma omp parallel for schedule(static) num_threads(threads) if (cn > 256)
            for (unsigned int k = 0; k < cn; ++k)
              if (cached[k] && !hit0[k].name) copy_template_info(cached[k]->proto, &hit0[k]);
            for (unsigned int k = 0; k < cn; ++k) hit0[k].entry = ent[k];
            timer.lap(PhaseTimer::PATHS, "prototype copies");
          }
        } else {
          for (unsigned int k = 0; k < cn; ++k) {
            copy_template_info(ret_hits[resident[ent[k]].first_hit], &hit0[k]);
            hit0[k].entry = ent[k];
          }
        }

        // ---- the ss mode of every SIMD batch of the reference (:14-22), then one launch per (set, mode) ----
        std::vector<const ResidentTemplate*> rt(cn);  // one table lookup per template and round
        for (unsigned int k = 0; k < cn; ++k) rt[k] = &resident[ent[k]];
        // The caller's pb as the reference leaves it with one thread: the background of the template it read LAST - every round
        // reads its templates again there (:144), in the order of the sorted block.  A raw template (cached, from the sidecar or
        // read in this round) carries pb0; behind a host-prepared one of round 0 pb still holds what its read left (later rounds do
        // not read again here: for an HMMER-format template at the end of a later round's block pb keeps the value of round 0).
        if (rt[cn - 1]->raw) memcpy(pb, pb0, sizeof(pb0));
        std::vector<int> batch_mode(cn);
        std::vector<uint8_t> shorter(cn, 0);  // shorter than the longest template of its batch (HMMSimd::L, src/hhhmmsimd.cpp:97)
        for (unsigned int b = 0; b < cn; b += VECSIZE_FLOAT) {
          int consensus = 0xFF, Lbatch = 0;
          const unsigned int e = imin(cn, b + VECSIZE_FLOAT);
          for (unsigned int k = b; k < e; ++k) {
            consensus &= rt[k]->ss_pair_mode;
            Lbatch = imax(Lbatch, rt[k]->L);
          }
          const int mode = select_ss_mode(consensus);
          for (unsigned int k = b; k < e; ++k) {
            batch_mode[k] = mode;
            shorter[k] = rt[k]->L < Lbatch;
            if (getenv("HHV_DROPIN_DEBUG") && strstr(ent[k]->getName(), getenv("HHV_DROPIN_DEBUG")))
              fprintf(stderr, "hhviterbirunner_hip debug: %s round %d position %u of %u: L %d, sequence_length %d, batch L %d\n",
                      ent[k]->getName(), alignment, c0 + k, m, rt[k]->L, ent[k]->sequence_length, Lbatch);
          }
        }
        std::map<std::pair<hhv_tset*, int>, std::vector<unsigned int> > groups;
        for (unsigned int k = 0; k < cn; ++k) groups[std::make_pair(rt[k]->set, batch_mode[k])].push_back(k);
        {
          // device section 2: alignment, backtrace, Hit scores, paths - the groups of one device one after the other, the
          // devices side by side (one host thread per context)
          std::lock_guard<std::mutex> lock(tc.device);
          search.install();
          const int n_slots = (int)tc.slots.size();
          typedef std::map<std::pair<hhv_tset*, int>, std::vector<unsigned int> >::iterator GroupIt;
          std::vector<std::vector<GroupIt> > groups_of_slot(n_slots);
          for (GroupIt g = groups.begin(); g != groups.end(); ++g) groups_of_slot[rt[g->second[0]]->dev].push_back(g);
          auto run_slot = [&](int d, bool timed) {
            for (size_t w = 0; w < groups_of_slot[d].size(); ++w) {
              GroupIt g = groups_of_slot[d][w];
              const std::vector<unsigned int>& mem = g->second;
              const int n = (int)mem.size();
              std::vector<int32_t> ids(n);
              std::vector<const ResidentTemplate*> tmpl(n);
              std::vector<Hit*> out(n);
              std::vector<uint8_t> not_longest(n);
              bool whole = (n == hhv_tset_size(g->first.first));
              for (int k = 0; k < n; ++k) {
                tmpl[k] = rt[mem[k]];
                ids[k] = tmpl[k]->index;
                out[k] = &hit0[mem[k]];
                not_longest[k] = shorter[mem[k]];
                whole = whole && ids[k] == k;
              }
              search.run(d, timed, g->first.first, whole ? NULL : ids.data(), n, g->first.second, tmpl, out, not_longest);
            }
          };
          if (n_slots == 1) {
            run_slot(0, true);
          } else {
            std::vector<std::thread> workers;
            for (int d = 0; d < n_slots; ++d)
              if (!groups_of_slot[d].empty()) workers.push_back(std::thread(run_slot, d, false));
            for (size_t w = 0; w < workers.size(); ++w) workers[w].join();
            timer.lap(PhaseTimer::ALIGN);
          }
        }
      }

      timer.lap(PhaseTimer::OTHER, "after run");
      // ---- merge_thread_results (:249-271) ----
      for (unsigned int k = 0; k < m; ++k) {
        Hit& h = ret_hits[first_hit_of_block + k];
        h.irep = (alignment + 1);
        if (h.score > par.smin) {
          next_work.push_back(h.entry);
          search.excludeAlignments[std::string(h.entry->getName())].push_back(first_hit_of_block + k);
        }
      }
      HH_LOG(INFO) << (block_start + m) << " alignments done" << std::endl;
      timer.lap(PhaseTimer::OTHER, "merge");

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

  timer.lap(PhaseTimer::OTHER, "early stop");
  {
    std::lock_guard<std::mutex> lock(tc.device);
    for (size_t k = 0; k < search_sets.size(); ++k) hhv_tset_free(search_sets[k]);
    if (device_prepare) tc.active--;
  }
  for (size_t k = 0; k < own_ss.size(); ++k) delete own_ss[k];
  for (int k = 0; k < threads; ++k) {
    delete t_simd[k];
    delete t_hmm[k];
    delete t_hdr[k];
  }
  timer.lap(PhaseTimer::OTHER, "free sets");
  std::vector<Hit> result;
  result.swap(ret_hits);
  return result;
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
