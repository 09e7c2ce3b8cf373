// oracle/ref_runner_harness.cpp -- TEST INFRASTRUCTURE (never linked into the product).
//
// Drives ViterbiRunner::alignment (src/hhviterbirunner.h:50-58) of hh-suite v3.3.0 exactly the way HHblits::run
// does (src/hhblits.cpp:1136-1215): substitution matrices, query read + PrepareQueryHMM, HMMSimd::MapOneHMM,
// ViterbiMatrix per thread, a std::vector<HHEntry*> of templates.  The templates are .hhm texts held in memory
// (MemEntry reads them through the reference's own HHEntry::getTemplateHMM(FILE*, ...), src/hhdatabase.cpp:398-462).
//
// The file is compiled TWICE by oracle/Makefile into oracle/_ref/libhhref_dropin.so:
//   - as is:            ref_runner_run_cpu  -> the reference's own src/hhviterbirunner.cpp (AVX2, OpenMP)
//   - -DHARNESS_HIP:    ref_runner_run_hip  -> hh-suite_amd/dropin/hhviterbirunner_hip.cpp, the drop-in replacement
//                       of that translation unit; class names are suffixed by macro so that both implementations
//                       of "ViterbiRunner" can live in one process and be compared hit by hit.
#ifdef HARNESS_HIP
#define ViterbiRunner ViterbiRunnerHip
#define ViterbiConsumerThread ViterbiConsumerThreadHip
#define RUN_NAME ref_runner_run_hip
#else
#define RUN_NAME ref_runner_run_cpu
#endif

#include <sys/time.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "hhdatabase.h"
#include "hhdecl.h"
#include "hhfunc.h"
#include "hhhmm.h"
#include "hhhmmsimd.h"
#include "hhmatrices.h"
#include "hhviterbimatrix.h"
#include "hhviterbirunner.h"

namespace {

class MemEntry : public HHEntry {
 public:
  MemEntry(int index, const char* name, const char* text, size_t len, int sequence_length)
      : HHEntry(sequence_length), index(index), name_(name), text_(text), len_(len) {}
  void getTemplateHMM(Parameters& par, char use_global_weights, const float qsc, int& format, float* pb,
                      const float S[20][20], const float Sim[20][20], HMM* t) {
    FILE* f = fmemopen((void*)text_, len_, "r");
    std::vector<char> nm(name_.begin(), name_.end());
    nm.push_back('\0');
    HHEntry::getTemplateHMM(f, nm.data(), par, use_global_weights, qsc, format, pb, S, Sim, t);
    fclose(f);
  }
  char* getName() { return const_cast<char*>(name_.c_str()); }
  int index;

 private:
  std::string name_;
  const char* text_;
  size_t len_;
};

}  // namespace

extern "C" {

struct rr_hit {
  int32_t entry, irep, lastrep, L, nsteps, matched_cols, i1, j1, i2, j2, ssm1, ssm2, n_display;
  float score, score_ss, score_aass, Neff_HMM;
  char name[64];
};

// opts_i: [0] loc [1] altali [2] ssm [3] early_stopping_filter [4] prefilter [5] dbsize [6] maxres [7] threads
//         [8] pc_hhm_nocontext_mode (-1 = default) [9] columnscore (-1 = default)
// opts_f: [0] smin [1] filter_thresh [2] egq [3] egt [4] ssw
// seq_len[k]: the length the cs219 entry would announce (HHEntry::sequence_length, the sort key of :117-119)
// alignment_seconds (nullable): wall time of the ViterbiRunner::alignment call alone.
// Returns the number of hits (may exceed cap_hits; only cap_hits are written) or a negative error.
int RUN_NAME(const char* query_hhm, size_t query_len, int n, const char* const* tmpl_hhm, const size_t* tmpl_len,
             const char* const* names, const int32_t* seq_len, const int32_t* opts_i, const float* opts_f,
             const char* exclstr, const char* template_exclstr, int cap_hits, rr_hit* hits, int path_cap, int32_t* pi,
             int32_t* pj, int8_t* pstates, float* pS, float* pS_ss, double* alignment_seconds) {
  Parameters par(0, NULL);
  Log::reporting_level() = WARNING;
  par.nocontxt = 1;  // the context_data.crf blob is not part of the reference tree (.MISSING_LARGE_BLOBS)
  par.loc = opts_i[0];
  par.altali = opts_i[1];
  par.ssm = opts_i[2];
  par.early_stopping_filter = opts_i[3] != 0;
  par.prefilter = opts_i[4] != 0;
  par.dbsize = opts_i[5];
  par.maxres = opts_i[6];
  par.threads = opts_i[7];
  if (opts_i[8] >= 0) par.pc_hhm_nocontext_mode = opts_i[8];
  if (opts_i[9] >= 0) par.columnscore = opts_i[9];
  par.smin = opts_f[0];
  par.filter_thresh = opts_f[1];
  par.egq = opts_f[2];
  par.egt = opts_f[3];
  par.ssw = opts_f[4];
  std::vector<char> ex, tex;
  if (exclstr && *exclstr) {
    ex.assign(exclstr, exclstr + strlen(exclstr) + 1);
    par.exclstr = ex.data();
  }
  if (template_exclstr && *template_exclstr) {
    tex.assign(template_exclstr, template_exclstr + strlen(template_exclstr) + 1);
    par.template_exclstr = tex.data();
  }

  float pb[21];
  float P[20][20], R[20][20], S[20][20], Sim[20][20];
  SetSubstitutionMatrix(par.matrix, pb, P, R, S, Sim);
  static float S73[NDSSP][NSSPRED][MAXCF], S33[NSSPRED][MAXCF][NSSPRED][MAXCF], S37[NSSPRED][MAXCF][NDSSP];
  SetSecStrucSubstitutionMatrix(par.ssa, S73, S37, S33);  // src/hhblits.cpp:32

  // query: HMM::Read + PrepareQueryHMM (src/hhblits.cpp:1090-1130 for an .hhm query)
  HMM* q = new HMM(MAXSEQDIS, par.maxres);
  {
    FILE* f = fmemopen((void*)query_hhm, query_len, "r");
    char path[NAMELEN] = "";
    if (!q->Read(f, par.maxcol, par.nseqdis, pb, path)) {
      fclose(f);
      return -1;
    }
    fclose(f);
  }
  char input_format = 0;
  PrepareQueryHMM(par, input_format, q, NULL, NULL, pb, R);
  HMMSimd q_vec(par.maxres);
  q_vec.MapOneHMM(q);

  std::vector<HHEntry*> entries;
  int maxL = 0;
  for (int k = 0; k < n; ++k) {
    entries.push_back(new MemEntry(k, names[k], tmpl_hhm[k], tmpl_len[k], seq_len[k]));
    maxL = std::max(maxL, seq_len[k]);
  }
  ViterbiMatrix** vm = new ViterbiMatrix*[par.threads];
  for (int t = 0; t < par.threads; ++t) {
    vm[t] = new ViterbiMatrix();
    vm[t]->AllocateBacktraceMatrix(q->L, std::min(maxL, par.maxres));  // src/hhblits.cpp:1196
  }
  std::vector<HHblitsDatabase*> dbs;
  ViterbiRunner runner(vm, dbs, par.threads);
  struct timeval t0, t1;
  gettimeofday(&t0, NULL);
  std::vector<Hit> res = runner.alignment(par, &q_vec, entries, par.qsc_db, pb, S, Sim, R, par.ssm, S73, S33, S37);
  gettimeofday(&t1, NULL);
  if (alignment_seconds) *alignment_seconds = (t1.tv_sec - t0.tv_sec) + 1e-6 * (t1.tv_usec - t0.tv_usec);

  const int m = (int)res.size();
  for (int h = 0; h < m; ++h) {
    Hit& x = res[h];
    if (h < cap_hits) {
      rr_hit o;
      memset(&o, 0, sizeof(o));
      o.entry = static_cast<MemEntry*>(x.entry)->index;
      o.irep = x.irep;
      o.lastrep = x.lastrep;
      o.L = x.L;
      o.nsteps = x.nsteps;
      o.matched_cols = x.matched_cols;
      o.i1 = x.i1;
      o.j1 = x.j1;
      o.i2 = x.i2;
      o.j2 = x.j2;
      o.ssm1 = x.ssm1;
      o.ssm2 = x.ssm2;
      o.n_display = x.n_display;
      o.score = x.score;
      o.score_ss = x.score_ss;
      o.score_aass = x.score_aass;
      o.Neff_HMM = x.Neff_HMM;
      strncpy(o.name, x.name, sizeof(o.name) - 1);
      hits[h] = o;
      const int c = std::min(path_cap, x.nsteps + 1);
      for (int s = 1; s < c; ++s) {
        const size_t at = (size_t)h * path_cap + s;
        pi[at] = x.i[s];
        pj[at] = x.j[s];
        pstates[at] = x.states[s];
        pS[at] = x.S[s];
        pS_ss[at] = x.S_ss[s];
      }
    }
    x.Delete();
  }
  for (int t = 0; t < par.threads; ++t) delete vm[t];
  delete[] vm;
  for (int k = 0; k < n; ++k) delete entries[k];
  delete q;
  return m;
}

}  // extern "C"
