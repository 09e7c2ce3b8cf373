// viterbi_runner.cpp -- see viterbi_runner.h.  Host glue only; every number comes out of the C ABI.
#include "viterbi_runner.h"

#include <string.h>

#include <algorithm>
#include <thread>

namespace hhv {

namespace {
const int VITERBI_PATH_WIDTH = 40;  // src/hhdecl.h:50

void check(int rc, const char* what) {
  if (rc != HHV_OK) throw Error(rc, std::string(what) + ": " + hhv_last_error());
}

struct CtxGuard {
  hhv_ctx* c = nullptr;
  ~CtxGuard() {
    if (c) hhv_destroy(c);
  }
};
struct SetGuard {
  hhv_tset* t = nullptr;
  ~SetGuard() {
    if (t) hhv_tset_free(t);
  }
};
}  // namespace

void ExcludeAlignment(std::vector<uint8_t>& mask, int Lq, int Lt, const int32_t* i_steps, const int32_t* j_steps,
                      int nsteps) {
  const size_t W = (size_t)Lt + 1;
  for (int step = 1; step < nsteps; ++step) {  // the last step is skipped, like the reference (:65)
    const int i = i_steps[step], j = j_steps[step];
    for (int ii = std::max(i - VITERBI_PATH_WIDTH, 1); ii <= std::min(i + VITERBI_PATH_WIDTH, Lq); ++ii)
      mask[(size_t)ii * W + j] = 1;
    for (int jj = std::max(j - VITERBI_PATH_WIDTH, 1); jj <= std::min(j + VITERBI_PATH_WIDTH, Lt); ++jj)
      mask[(size_t)i * W + jj] = 1;
  }
}

namespace {
// strint (src/util.cpp:133-151): next integer in the string, a leading '-' negates; -> false when none is left
bool next_int(const std::string& s, size_t& pos, int& out) {
  while (pos < s.size() && !(s[pos] >= '0' && s[pos] <= '9')) ++pos;
  if (pos >= s.size()) return false;
  const bool neg = pos > 0 && s[pos - 1] == '-';
  long v = 0;
  while (pos < s.size() && s[pos] >= '0' && s[pos] <= '9') v = v * 10 + (s[pos++] - '0');
  out = (int)(neg ? -v : v);
  return true;
}
}  // namespace

std::vector<int32_t> ParseRegions(const std::string& exclstr) {
  std::vector<int32_t> pairs;
  size_t pos = 0;
  int a, b;
  while (next_int(exclstr, pos, a) && next_int(exclstr, pos, b)) {
    pairs.push_back(std::max(1, std::abs(a)));
    pairs.push_back(std::abs(b));
  }
  return pairs;
}

void ExcludeRegions(std::vector<uint8_t>& mask, int Lq, int Lt, const std::string& exclstr) {
  size_t pos = 0;
  int a, b;
  while (next_int(exclstr, pos, a) && next_int(exclstr, pos, b)) {
    const int i0 = std::max(1, std::abs(a)), i1 = std::min(std::abs(b), Lq);
    for (int i = i0; i <= i1; ++i)
      for (int j = 1; j <= Lt; ++j) mask[(size_t)i * (Lt + 1) + j] = 1;
  }
}

void ExcludeTemplateRegions(std::vector<uint8_t>& mask, int Lq, int Lt, const std::string& exclstr) {
  size_t pos = 0;
  int a, b;
  while (next_int(exclstr, pos, a) && next_int(exclstr, pos, b)) {
    const int j0 = std::max(1, std::abs(a)), j1 = std::min(std::abs(b), Lt);
    for (int j = j0; j <= j1; ++j)
      for (int i = 1; i <= Lq; ++i) mask[(size_t)i * (Lt + 1) + j] = 1;
  }
}

std::vector<Hit> ViterbiRunner::alignment(const Parameters& par, const Profile& q,
                                          const std::vector<Profile>& templates) {
  std::vector<Hit> ret_hits;
  const int n = (int)templates.size();
  if (n == 0) return ret_hits;
  if (!q.p || !q.tr || q.L < 1) throw Error(HHV_E_ARG, "ViterbiRunner::alignment: empty query");

  hhv_params hp;
  hp.device = device_;
  hp.local = par.loc;
  hp.egq = par.egq;
  hp.egt = par.egt;
  hp.shift = par.shift;
  hp.corr = par.corr;
  hp.ssw = par.ssw;
  hp.ss_mode = par.ssm;
  CtxGuard ctx;
  check(hhv_create(&ctx.c, &hp), "hhv_create");
  check(hhv_set_query(ctx.c, q.p, q.tr, q.L), "hhv_set_query");

  // the database block stays resident for all alternative-alignment rounds
  std::vector<int32_t> L(n);
  std::vector<const float*> pp(n), tt(n);
  for (int k = 0; k < n; ++k) {
    L[k] = templates[k].L;
    pp[k] = templates[k].p;
    tt[k] = templates[k].tr;
  }
  SetGuard all;
  check(hhv_upload_templates(ctx.c, n, L.data(), pp.data(), tt.data(), &all.t), "hhv_upload_templates");

  // excludeAlignments (src/hhviterbirunner.cpp:100,262-268): accumulated mask per template
  std::vector<int> to_align(n);
  for (int k = 0; k < n; ++k) to_align[k] = k;
  // -excl / -template_excl regions are masked in every round, including the first (:157-164); the masks are built on
  // the device (hhv_set_celloff_paths) from the ranges and, in later rounds, the paths of the earlier alignments
  const bool regions = !par.exclstr.empty() || !par.template_exclstr.empty();
  const std::vector<int32_t> qr = ParseRegions(par.exclstr), tr = ParseRegions(par.template_exclstr);
  if (regions)
    check(hhv_set_celloff_paths(ctx.c, all.t, 0, nullptr, nullptr, nullptr, nullptr, (int32_t)qr.size() / 2, qr.data(),
                                (int32_t)tr.size() / 2, tr.data()),
          "hhv_set_celloff_paths");
  // earlier alignments of every template (entries 1..nsteps of the path), accumulated over the rounds (:273-289)
  std::vector<std::vector<int32_t> > prev_i(n), prev_j(n);
  std::vector<std::vector<int64_t> > prev_off(n);

  for (int alignment = 0; alignment < par.altali && !to_align.empty(); ++alignment) {
    // round 0 runs on the resident set; later rounds on the (usually much smaller) surviving subset
    SetGuard sub;
    hhv_tset* ts = all.t;
    const int m = (int)to_align.size();
    if (alignment > 0) {
      // the surviving templates are copied on the device from the resident set: nothing is packed or uploaded again
      std::vector<int32_t> ids(to_align.begin(), to_align.end());
      check(hhv_tset_gather(ctx.c, all.t, ids.data(), m, &sub.t), "hhv_tset_gather");
      ts = sub.t;
      std::vector<int32_t> template_of, pi, pj;
      std::vector<int64_t> poff(1, 0);
      for (int t = 0; t < m; ++t) {
        const int k = to_align[t];
        for (size_t a = 0; a + 1 < prev_off[k].size(); ++a) {
          template_of.push_back(t);
          pi.insert(pi.end(), prev_i[k].begin() + prev_off[k][a], prev_i[k].begin() + prev_off[k][a + 1]);
          pj.insert(pj.end(), prev_j[k].begin() + prev_off[k][a], prev_j[k].begin() + prev_off[k][a + 1]);
          poff.push_back((int64_t)pi.size());
        }
      }
      check(hhv_set_celloff_paths(ctx.c, ts, (int32_t)template_of.size(), template_of.data(), poff.data(), pi.data(), pj.data(),
                                  (int32_t)qr.size() / 2, qr.data(), (int32_t)tr.size() / 2, tr.data()),
            "hhv_set_celloff_paths");
    }
    std::vector<hhv_hit> hits(m);
    check(hhv_align(ctx.c, ts, (alignment > 0 || regions) ? HHV_ALIGN_CELLOFF : HHV_ALIGN_BACKTRACE, nullptr),
          "hhv_align");
    check(hhv_hits(ctx.c, ts, hits.data()), "hhv_hits");

    std::vector<int> next;
    for (int t = 0; t < m; ++t) {
      const int k = to_align[t];
      const hhv_hit& h = hits[t];
      Hit hit;
      hit.entry = k;
      hit.irep = alignment + 1;                       // :257
      hit.lastrep = (h.score <= par.smin) ? 1 : 0;    // :37
      hit.score = h.score;
      hit.score_ss = h.score_ss;
      hit.score_aass = -h.score;                      // hhviterbi.cpp:252
      hit.i1 = h.i1;
      hit.j1 = h.j1;
      hit.i2 = h.i2;
      hit.j2 = h.j2;
      hit.nsteps = h.nsteps;
      hit.matched_cols = h.matched_cols;
      const int cap = h.nsteps + 1;
      hit.i.resize(cap);
      hit.j.resize(cap);
      hit.states.resize(cap);
      hit.S.resize(cap);
      hit.S_ss.assign(cap, 0.0f);
      int32_t ns = 0;
      check(hhv_hit_path(ctx.c, ts, t, cap, hit.i.data(), hit.j.data(), hit.states.data(), hit.S.data(), &ns),
            "hhv_hit_path");
      if (h.score > par.smin) {                       // :260-268
        next.push_back(k);
        if (prev_off[k].empty()) prev_off[k].push_back(0);
        prev_i[k].insert(prev_i[k].end(), hit.i.begin() + 1, hit.i.end());   // entries 1..nsteps
        prev_j[k].insert(prev_j[k].end(), hit.j.begin() + 1, hit.j.end());
        prev_off[k].push_back((int64_t)prev_i[k].size());
      }
      ret_hits.push_back(std::move(hit));
    }
    to_align.swap(next);
  }
  return ret_hits;
}

ShardedViterbiRunner::ShardedViterbiRunner(const std::vector<int>& devices) : devices_(devices) {
  if (devices_.empty()) {
    int32_t n = 0;
    check(hhv_device_count(&n), "hhv_device_count");
    for (int d = 0; d < n; ++d) devices_.push_back(d);
  }
}

std::vector<Hit> ShardedViterbiRunner::alignment(const Parameters& par, const Profile& q,
                                                 const std::vector<Profile>& templates) {
  const int n = (int)templates.size(), S = (int)devices_.size();
  shard_sizes_.assign(S, 0);
  if (n == 0) return std::vector<Hit>();
  if (S == 1) {
    shard_sizes_[0] = n;
    return ViterbiRunner(devices_[0]).alignment(par, q, templates);
  }
  std::vector<int32_t> L(n), shard_of(n);
  for (int k = 0; k < n; ++k) L[k] = templates[k].L;
  check(hhv_shard_plan(n, L.data(), S, shard_of.data()), "hhv_shard_plan");
  // the templates of a shard keep their input order, so that a shard's own hit order is the global one restricted to it
  std::vector<std::vector<Profile> > part(S);
  std::vector<std::vector<int> > global_id(S);
  for (int k = 0; k < n; ++k) {
    part[shard_of[k]].push_back(templates[k]);
    global_id[shard_of[k]].push_back(k);
  }
  std::vector<std::vector<Hit> > hits(S);
  std::vector<int> status(S, HHV_OK);
  std::vector<std::string> message(S);
  std::vector<std::thread> workers;
  for (int s = 0; s < S; ++s) {
    shard_sizes_[s] = (int)part[s].size();
    workers.push_back(std::thread([&, s]() {   // one host thread per context, as the C ABI asks
      try {
        if (!part[s].empty()) hits[s] = ViterbiRunner(devices_[s]).alignment(par, q, part[s]);
      } catch (const Error& e) {
        status[s] = e.status;
        message[s] = e.what();
      } catch (const std::exception& e) {
        status[s] = HHV_E_MEMORY;
        message[s] = e.what();
      }
    }));
  }
  for (size_t w = 0; w < workers.size(); ++w) workers[w].join();
  for (int s = 0; s < S; ++s)
    if (status[s] != HHV_OK) throw Error(status[s], "shard " + std::to_string(s) + ": " + message[s]);
  // merge in hit order: a single runner returns round 1 of every template in template order, then round 2, ...
  size_t total = 0;
  for (int s = 0; s < S; ++s) {
    for (size_t h = 0; h < hits[s].size(); ++h) hits[s][h].entry = global_id[s][hits[s][h].entry];
    total += hits[s].size();
  }
  std::vector<Hit> merged;
  merged.reserve(total);
  std::vector<size_t> at(S, 0);
  while (merged.size() < total) {  // S-way merge of lists that are each sorted by (irep, entry)
    int best = -1;
    for (int s = 0; s < S; ++s) {
      if (at[s] >= hits[s].size()) continue;
      const Hit& a = hits[s][at[s]];
      if (best < 0) {
        best = s;
        continue;
      }
      const Hit& b = hits[best][at[best]];
      if (a.irep < b.irep || (a.irep == b.irep && a.entry < b.entry)) best = s;
    }
    merged.push_back(std::move(hits[best][at[best]++]));
  }
  return merged;
}

}  // namespace hhv

// ---- C shim so that the parity tests (ctypes) can drive the C++ class ------------------------------
extern "C" {

struct hhvr_hit {
  int32_t entry, irep, lastrep;
  float score;
  int32_t i1, j1, i2, j2, nsteps, matched_cols;
};

// Runs hhv::ViterbiRunner::alignment.  hits_out: cap_hits records; path arrays: per hit `path_cap`
// entries at offset h*path_cap.  Returns the number of hits or a negative hhv_status.
int hhvr_alignment(int device, int loc, float egq, float egt, float shift, float corr, float ssw, int ssm, int altali,
                   float smin, const char* exclstr, const char* template_exclstr, const float* qp, const float* qtr, int Lq, int n, const int32_t* L,
                   const float* const* p, const float* const* tr, hhvr_hit* hits_out, int cap_hits, int path_cap,
                   int32_t* i_steps, int32_t* j_steps, int8_t* states, float* S) {
  try {
    hhv::Parameters par;
    par.loc = loc;
    par.egq = egq;
    par.egt = egt;
    par.shift = shift;
    par.corr = corr;
    par.ssw = ssw;
    par.ssm = ssm;
    par.altali = altali;
    par.smin = smin;
    if (exclstr) par.exclstr = exclstr;
    if (template_exclstr) par.template_exclstr = template_exclstr;
    hhv::Profile q;
    q.L = Lq;
    q.p = qp;
    q.tr = qtr;
    std::vector<hhv::Profile> ts(n);
    for (int k = 0; k < n; ++k) {
      ts[k].L = L[k];
      ts[k].p = p[k];
      ts[k].tr = tr[k];
    }
    // device >= 0: one GPU; device < 0: -device logical shards, shard s on device s % (number of GPUs) - the sharded runner
    std::vector<hhv::Hit> hits;
    if (device >= 0) {
      hhv::ViterbiRunner runner(device);
      hits = runner.alignment(par, q, ts);
    } else {
      int32_t ndev = 0;
      if (hhv_device_count(&ndev) != HHV_OK) return HHV_E_DEVICE;
      std::vector<int> devs;
      for (int s = 0; s < -device; ++s) devs.push_back(s % ndev);
      hhv::ShardedViterbiRunner runner(devs);
      hits = runner.alignment(par, q, ts);
    }
    const int m = (int)hits.size();
    for (int h = 0; h < m && h < cap_hits; ++h) {
      const hhv::Hit& x = hits[h];
      hhvr_hit o = {x.entry, x.irep, x.lastrep, x.score, x.i1, x.j1, x.i2, x.j2, x.nsteps, x.matched_cols};
      hits_out[h] = o;
      const int c = std::min(path_cap, x.nsteps + 1);
      if (i_steps) memcpy(i_steps + (size_t)h * path_cap, x.i.data(), c * sizeof(int32_t));
      if (j_steps) memcpy(j_steps + (size_t)h * path_cap, x.j.data(), c * sizeof(int32_t));
      if (states) memcpy(states + (size_t)h * path_cap, x.states.data(), c);
      if (S) memcpy(S + (size_t)h * path_cap, x.S.data(), c * sizeof(float));
    }
    return m;
  } catch (const hhv::Error& e) {
    return e.status;
  } catch (...) {
    return HHV_E_MEMORY;
  }
}

}  // extern "C"
