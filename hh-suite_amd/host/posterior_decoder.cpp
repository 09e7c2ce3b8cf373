// posterior_decoder.cpp -- see posterior_decoder.h.  Host-side batching and mask construction of the MAC realignment.
#include "posterior_decoder.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <map>

namespace hhv {

// wall-clock split of the last executeComputation: masks (host), hhv_mac_realign (staging + kernels + copies), paths
static double g_mac_ms[3] = {0, 0, 0};
static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

namespace {
enum { M2M = 0, M2I = 1, M2D = 2, I2M = 3, I2I = 4, D2M = 5, D2D = 6 };  // src/hhdecl.h:68
const int kPathWidth = 40;                                               // FWD_BKW_PATHWITDH, src/hhdecl.h:37
}  // namespace

void LinearTransitions(const float* tr, int L, bool is_query, float* out) {
  for (int i = 0; i <= L; ++i)
    for (int a = 0; a < 7; ++a) out[(size_t)i * 7 + a] = powf(2.0f, 1.0f * tr[(size_t)i * 7 + a]);
  float* t0 = out;
  float* tL = out + (size_t)L * 7;
  if (is_query) {
    t0[M2D] = t0[M2I] = 0.0f;
    t0[I2M] = t0[I2I] = 0.0f;
    t0[D2M] = t0[D2D] = 0.0f;
  } else {
    t0[M2M] = 1.0f;
    t0[M2D] = t0[M2I] = 0.0f;
    t0[I2M] = t0[I2I] = 0.0f;
    t0[D2M] = t0[D2D] = 0.0f;
  }
  tL[M2M] = 1.0f;
  tL[M2D] = tL[M2I] = 0.0f;
  tL[I2M] = tL[I2I] = 0.0f;
  tL[D2M] = 1.0f;
  tL[D2D] = 0.0f;
}

void MacCellOff(int Lq, int Lt, const MacParameters& par, const MacInput& hit, const std::vector<const MacAlignment*>& earlier,
                std::vector<uint8_t>* mask_out) {
  std::vector<uint8_t>& mask = *mask_out;
  const int pitch = Lt + 1;
  mask.assign((size_t)(Lq + 1) * pitch, 0);
  // Viterbi::InitializeForAlignment for two different HMMs (src/hhviterbi.cpp:337-357): minimum-overlap corners.
  // (maskViterbiAlignment below overwrites every cell, so this has no lasting effect - kept in the reference's order.)
  const int lmin = std::min(Lq, Lt);
  const int mo = par.min_overlap == 0 ? std::min(60, (int)(0.333f * lmin) + 1) : std::min(par.min_overlap, (int)(0.8f * lmin));
  for (int i = 0; i < mo; ++i)
    for (int j = std::max(0, i - mo + Lt + 1); j <= Lt; ++j) mask[(size_t)i * pitch + j] = 1;
  for (int i = std::max(0, Lq - mo + 1); i <= Lq; ++i)
    for (int j = 1; j < i + mo - Lq && j <= Lt; ++j) mask[(size_t)i * pitch + j] = 1;
  // maskViterbiAlignment (src/hhposteriordecoder.cpp:205-240): everything off except the two corner rectangles, then
  // a band of +-40 rows / columns around every step of the Viterbi path switched on
  for (int i = 1; i <= Lq; ++i) {
    // mask(i,j) = !((i < i1 && j < j1) || (i > i2 && j > j2)), written as row fills
    uint8_t* row = &mask[(size_t)i * pitch];
    memset(row + 1, 1, (size_t)Lt);
    if (i < hit.i1 && hit.j1 > 1) memset(row + 1, 0, (size_t)std::min(hit.j1 - 1, Lt));
    if (i > hit.i2 && hit.j2 < Lt) memset(row + std::max(hit.j2 + 1, 1), 0, (size_t)(Lt - std::max(hit.j2 + 1, 1) + 1));
  }
  for (int step = hit.nsteps; step >= 1; --step)
    for (int i = std::max(1, hit.i[step] - kPathWidth); i <= std::min(Lq, hit.i[step] + kPathWidth); ++i)
      mask[(size_t)i * pitch + hit.j[step]] = 0;
  for (int step = hit.nsteps; step >= 1; --step)
    for (int j = std::max(1, hit.j[step] - kPathWidth); j <= std::min(Lt, hit.j[step] + kPathWidth); ++j)
      mask[(size_t)hit.i[step] * pitch + j] = 0;
  // excludeMACAlignment (:245-262) for the alignments found in earlier rounds
  for (size_t e = 0; e < earlier.size(); ++e) {
    const MacAlignment& al = *earlier[e];
    const int first = al.nsteps == 0 ? 0 : 1;  // alt_i holds the single start cell when the backtrace did not start in MM
    for (int s = first; s <= al.nsteps; ++s) {
      const int i = al.i[s], j = al.j[s];
      for (int ii = std::max(i - 2, 1); ii <= std::min(i + 2, Lq); ++ii) mask[(size_t)ii * pitch + j] = 1;
      for (int jj = std::max(j - 2, 1); jj <= std::min(j + 2, Lt); ++jj) mask[(size_t)i * pitch + jj] = 1;
    }
  }
  // exclude_regions / exclude_template_regions (:121-149) - same parsing as the Viterbi stage
  if (!par.exclstr.empty()) ExcludeRegions(mask, Lq, Lt, par.exclstr);
  if (!par.template_exclstr.empty()) ExcludeTemplateRegions(mask, Lq, Lt, par.template_exclstr);
  for (int j = 0; j <= Lt; ++j) mask[j] = 0;  // row / column 0 are never read
  for (int i = 0; i <= Lq; ++i) mask[(size_t)i * pitch] = 0;
}

std::vector<MacAlignment> PosteriorDecoderRunner::executeComputation(const MacParameters& par, const Profile& q,
                                                                     const std::vector<Profile>& templates,
                                                                     const std::vector<MacInput>& hits) {
  std::vector<MacAlignment> out(hits.size());
  // group by template, each group ordered by irep (:54-66; ties keep the input order)
  std::map<int, std::vector<int> > groups;
  for (size_t h = 0; h < hits.size(); ++h) {
    if (hits[h].entry < 0 || hits[h].entry >= (int)templates.size()) throw Error(HHV_E_ARG, "MacInput.entry out of range");
    groups[hits[h].entry].push_back((int)h);
  }
  size_t rounds = 0;
  for (std::map<int, std::vector<int> >::iterator g = groups.begin(); g != groups.end(); ++g) {
    std::stable_sort(g->second.begin(), g->second.end(), [&](int a, int b) { return hits[a].irep < hits[b].irep; });
    rounds = std::max(rounds, g->second.size());
  }
  g_mac_ms[0] = g_mac_ms[1] = g_mac_ms[2] = 0;
  for (size_t r = 0; r < rounds; ++r) {
    const double t_start = now_ms();
    std::vector<int> batch;  // hit indices of this round
    for (std::map<int, std::vector<int> >::iterator g = groups.begin(); g != groups.end(); ++g)
      if (g->second.size() > r) batch.push_back(g->second[r]);
    const int n = (int)batch.size();
    // what the device needs to build the masks realign() builds (MacCellOff above is the host statement of the same)
    std::vector<hhv_mac_input> in(n);
    std::vector<std::vector<int32_t> > ex_i(n), ex_j(n);
    std::vector<const float*> tp(n), ttr(n);
    std::vector<int32_t> Lt(n);
    for (int b = 0; b < n; ++b) {
      const MacInput& hit = hits[batch[b]];
      const Profile& t = templates[hit.entry];
      const std::vector<int>& grp = groups[hit.entry];
      for (size_t e = 0; e < r; ++e) {
        const MacAlignment& al = out[grp[e]];
        for (int s = (al.nsteps == 0 ? 0 : 1); s <= al.nsteps; ++s) {
          ex_i[b].push_back(al.i[s]);
          ex_j[b].push_back(al.j[s]);
        }
      }
      in[b].i1 = hit.i1;
      in[b].j1 = hit.j1;
      in[b].i2 = hit.i2;
      in[b].j2 = hit.j2;
      in[b].nsteps = hit.nsteps;
      in[b].i = hit.i;
      in[b].j = hit.j;
      in[b].n_excluded = (int32_t)ex_i[b].size();
      in[b].excluded_i = ex_i[b].data();
      in[b].excluded_j = ex_j[b].data();
      tp[b] = t.p;
      ttr[b] = t.tr;
      Lt[b] = t.L;
    }
    const std::vector<int32_t> qr = ParseRegions(par.exclstr), tr = ParseRegions(par.template_exclstr);
    const double t_masks = now_ms();
    hhv_macset* ms = nullptr;
    std::vector<hhv_mac_hit> res(n);
    int rc;
    if (resident_) {
      std::vector<int32_t> template_of(n);
      for (int b = 0; b < n; ++b) template_of[b] = hits[batch[b]].resident >= 0 ? hits[batch[b]].resident : hits[batch[b]].entry;
      rc = hhv_mac_realign_tset(ctx_, q.p, q.tr, q.L, resident_, n, template_of.data(), ttr.data(), in.data(), (int32_t)qr.size() / 2,
                                qr.data(), (int32_t)tr.size() / 2, tr.data(), par.loc, par.shift, par.mact, &ms, res.data());
    } else {
      rc = hhv_mac_realign_hits(ctx_, q.p, q.tr, q.L, n, Lt.data(), tp.data(), ttr.data(), in.data(), (int32_t)qr.size() / 2,
                                qr.data(), (int32_t)tr.size() / 2, tr.data(), par.loc, par.shift, par.mact, &ms, res.data());
    }
    if (rc != HHV_OK) throw Error(rc, hhv_last_error());
    const double t_dp = now_ms();
    for (int b = 0; b < n; ++b) {
      MacAlignment& al = out[batch[b]];
      al.entry = hits[batch[b]].entry;
      al.irep = hits[batch[b]].irep;
      al.Pforward = res[b].Pforward;
      al.sum_of_probs = res[b].sum_of_probs;
      al.i1 = res[b].i1;
      al.j1 = res[b].j1;
      al.i2 = res[b].i2;
      al.j2 = res[b].j2;
      al.nsteps = res[b].nsteps;
      al.matched_cols = res[b].matched_cols;
      const int cap = al.nsteps + 1;
      al.i.assign(cap, 0);
      al.j.assign(cap, 0);
      al.states.assign(cap, 0);
      al.S.assign(cap, 0.f);
      al.P_posterior.assign(cap, 0.f);
      int32_t ns = 0;
      rc = hhv_mac_path(ms, b, cap, al.i.data(), al.j.data(), al.states.data(), al.S.data(), al.P_posterior.data(), &ns);
      if (rc != HHV_OK) {
        hhv_macset_free(ms);
        throw Error(rc, hhv_last_error());
      }
    }
    hhv_macset_free(ms);
    g_mac_ms[0] += t_masks - t_start;
    g_mac_ms[1] += t_dp - t_masks;
    g_mac_ms[2] += now_ms() - t_dp;
  }
  return out;
}

}  // namespace hhv

extern "C" {

void hhvr_mac_last_timing(double* ms3) { memcpy(ms3, hhv::g_mac_ms, sizeof(hhv::g_mac_ms)); }

void hhvr_linear_transitions(const float* tr_log2, int32_t L, int32_t is_query, float* out) {
  hhv::LinearTransitions(tr_log2, L, is_query != 0, out);
}

int hhvr_mac_celloff(int32_t Lq, int32_t Lt, int32_t min_overlap, const char* exclstr, const char* template_exclstr,
                     const int32_t* hit_row, const int32_t* path_i, const int32_t* path_j, int32_t n_prev,
                     const int32_t* prev_off, const int32_t* prev_i, const int32_t* prev_j, uint8_t* mask) {
  hhv::MacParameters par;
  par.min_overlap = min_overlap;
  if (exclstr) par.exclstr = exclstr;
  if (template_exclstr) par.template_exclstr = template_exclstr;
  hhv::MacInput hit;
  hit.entry = hit_row[0];
  hit.irep = hit_row[1];
  hit.i1 = hit_row[2];
  hit.j1 = hit_row[3];
  hit.i2 = hit_row[4];
  hit.j2 = hit_row[5];
  hit.nsteps = hit_row[6];
  hit.i = path_i;
  hit.j = path_j;
  std::vector<hhv::MacAlignment> prev(n_prev);
  std::vector<const hhv::MacAlignment*> earlier;
  for (int k = 0; k < n_prev; ++k) {
    // prev lists are alt_i/alt_j (no unused entry 0): rebuild the 1-based arrays MacCellOff walks
    const int cnt = prev_off[k + 1] - prev_off[k];
    prev[k].nsteps = cnt;
    prev[k].i.assign(cnt + 1, 0);
    prev[k].j.assign(cnt + 1, 0);
    for (int s = 0; s < cnt; ++s) {
      prev[k].i[s + 1] = prev_i[prev_off[k] + s];
      prev[k].j[s + 1] = prev_j[prev_off[k] + s];
    }
    earlier.push_back(&prev[k]);
  }
  std::vector<uint8_t> m;
  hhv::MacCellOff(Lq, Lt, par, hit, earlier, &m);
  memcpy(mask, m.data(), m.size());
  return 0;
}

int hhvr_mac_realign(hhv_ctx* ctx, hhv_tset* resident, int32_t loc, float shift, float mact, int32_t min_overlap, const char* exclstr,
                     const char* template_exclstr, const float* q_p, const float* q_tr_lin, int32_t Lq, int32_t n_templates,
                     const int32_t* Lt, const float* const* t_p, const float* const* t_tr_lin, int32_t n_hits,
                     const int32_t* hit_rows, const int64_t* path_off, const int32_t* path_i, const int32_t* path_j,
                     int32_t* out_scalars, double* out_real, int32_t pcap, int32_t* out_i, int32_t* out_j, int8_t* out_states,
                     float* out_S, float* out_P) {
  try {
    hhv::MacParameters par;
    par.loc = loc;
    par.shift = shift;
    par.mact = mact;
    par.min_overlap = min_overlap;
    if (exclstr) par.exclstr = exclstr;
    if (template_exclstr) par.template_exclstr = template_exclstr;
    hhv::Profile q;
    q.L = Lq;
    q.p = q_p;
    q.tr = q_tr_lin;
    std::vector<hhv::Profile> ts(n_templates);
    for (int k = 0; k < n_templates; ++k) {
      ts[k].L = Lt[k];
      ts[k].p = t_p ? t_p[k] : nullptr;
      ts[k].tr = t_tr_lin[k];
    }
    std::vector<hhv::MacInput> hits(n_hits);
    for (int h = 0; h < n_hits; ++h) {
      const int32_t* r = hit_rows + (size_t)h * 8;
      hits[h].resident = r[7];
      hits[h].entry = r[0];
      hits[h].irep = r[1];
      hits[h].i1 = r[2];
      hits[h].j1 = r[3];
      hits[h].i2 = r[4];
      hits[h].j2 = r[5];
      hits[h].nsteps = r[6] < 0 ? 0 : r[6];
      // nsteps < 0: the Viterbi alignment of this hit is the one the resident set holds (no path handed over)
      hits[h].i = r[6] < 0 ? nullptr : path_i + path_off[h];
      hits[h].j = r[6] < 0 ? nullptr : path_j + path_off[h];
    }
    hhv::PosteriorDecoderRunner runner(ctx);
    if (resident) runner.useResidentSet(resident);
    const std::vector<hhv::MacAlignment> res = runner.executeComputation(par, q, ts, hits);
    for (int h = 0; h < n_hits; ++h) {
      const hhv::MacAlignment& al = res[h];
      int32_t* sc = out_scalars + (size_t)h * 6;
      sc[0] = al.nsteps;
      sc[1] = al.i1;
      sc[2] = al.j1;
      sc[3] = al.i2;
      sc[4] = al.j2;
      sc[5] = al.matched_cols;
      out_real[(size_t)h * 2] = al.Pforward;
      out_real[(size_t)h * 2 + 1] = al.sum_of_probs;
      if (al.nsteps + 1 > pcap) return HHV_E_ARG;
      for (int s = 0; s <= al.nsteps; ++s) {
        out_i[(size_t)h * pcap + s] = al.i[s];
        out_j[(size_t)h * pcap + s] = al.j[s];
        out_states[(size_t)h * pcap + s] = al.states[s];
        out_S[(size_t)h * pcap + s] = al.S[s];
        out_P[(size_t)h * pcap + s] = al.P_posterior[s];
      }
    }
    return n_hits;
  } catch (const hhv::Error& e) {
    return e.status;
  }
}

}  // extern "C"
