// examples/search_example.cpp -- the host classes used the way hhsearch / hhblits would use them, from plain C++
// (no Python, no torch): hhv::ViterbiRunner::alignment -> hhv::PosteriorDecoderRunner::executeComputation on synthetic
// profiles.  Build: make example   (g++ + libhhv_runner.so + libhhviterbi_hip.so).  Prints one line per hit.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../hh-suite_amd/host/posterior_decoder.h"
#include "../hh-suite_amd/host/viterbi_runner.h"

namespace {
struct Rng {  // splitmix64: reproducible without <random>
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  double u() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return (double)((z ^ (z >> 31)) >> 11) / 9007199254740992.0;
  }
};
const float kPb[20] = {0.0787f, 0.0512f, 0.0448f, 0.0536f, 0.0135f, 0.0403f, 0.0610f, 0.0688f, 0.0229f, 0.0590f,
                       0.0964f, 0.0593f, 0.0237f, 0.0396f, 0.0483f, 0.0683f, 0.0585f, 0.0132f, 0.0321f, 0.0668f};

struct Hmm {
  int L;
  std::vector<float> f, p, tr;  // f: column probabilities, p: what the aligner sees, tr: log2 transitions
};

// random column distributions; if `from` is given, a noisy copy of a window of it (a homolog)
Hmm make_hmm(Rng& rng, int L, bool is_query, const Hmm* from) {
  Hmm h;
  h.L = L;
  h.f.assign((size_t)(L + 1) * 20, 0.f);
  h.p.assign((size_t)(L + 1) * 20, 0.f);
  h.tr.assign((size_t)(L + 1) * 7, 0.f);
  for (int i = 1; i <= L; ++i) {
    double col[20], sum = 0;
    for (int a = 0; a < 20; ++a) sum += (col[a] = pow(rng.u(), 6.0) + 1e-9);
    for (int a = 0; a < 20; ++a) {
      double v = 0.7 * col[a] / sum + 0.3 * kPb[a];
      if (from && i <= from->L) v = 0.75 * from->f[(size_t)i * 20 + a] + 0.25 * v;
      h.f[(size_t)i * 20 + a] = (float)v;
      h.p[(size_t)i * 20 + a] = is_query ? (float)v : (float)(v / kPb[a]);  // templates: null model folded in
    }
  }
  for (int i = 0; i <= L; ++i) {
    const double pI = 0.01 + 0.04 * rng.u(), pD = 0.01 + 0.04 * rng.u(), pII = 0.25 + 0.3 * rng.u(), pDD = 0.25 + 0.3 * rng.u();
    float* t = &h.tr[(size_t)i * 7];
    t[0] = (float)log2(1 - pI - pD);  // M2M M2I M2D I2M I2I D2M D2D (src/hhdecl.h:68)
    t[1] = (float)(0.6 * log2(pI));
    t[2] = (float)(0.6 * log2(pD));
    t[3] = (float)log2(1 - pII);
    t[4] = (float)(0.6 * log2(pII));
    t[5] = (float)log2(1 - pDD);
    t[6] = (float)(0.6 * log2(pDD));
    if (i == 0 || i == L) {
      t[0] = 0.f;
      t[1] = t[2] = -100000.f;
    }
    if (i == L) {
      t[6] = -100000.f;
      t[5] = 0.f;
    }
  }
  return h;
}
}  // namespace

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 64, Lq = 150;
  Rng rng(2024);
  const Hmm q = make_hmm(rng, Lq, true, nullptr);
  std::vector<Hmm> db;
  for (int k = 0; k < n; ++k) db.push_back(make_hmm(rng, 60 + (int)(rng.u() * 200), false, k % 4 == 0 ? &q : nullptr));

  try {
    // ---- Viterbi stage: ViterbiRunner::alignment (src/hhviterbirunner.cpp:75-210)
    hhv::Parameters par;
    par.loc = 1;
    par.ssm = 0;
    par.altali = 2;
    hhv::Profile Q;
    Q.L = q.L;
    Q.p = q.p.data();
    Q.tr = q.tr.data();
    std::vector<hhv::Profile> T(n);
    for (int k = 0; k < n; ++k) {
      T[k].L = db[k].L;
      T[k].p = db[k].p.data();
      T[k].tr = db[k].tr.data();
    }
    hhv::ViterbiRunner runner(0);
    const std::vector<hhv::Hit> hits = runner.alignment(par, Q, T);

    // ---- realign stage: PosteriorDecoderRunner::executeComputation (src/hhposteriordecoderrunner.cpp:43-119) for the
    // hits above a score threshold; linear transitions as the realign stage prepares them
    std::vector<float> q_lin((size_t)(q.L + 1) * 7);
    hhv::LinearTransitions(q.tr.data(), q.L, true, q_lin.data());
    hhv::Profile Qlin = Q;
    Qlin.tr = q_lin.data();
    std::vector<std::vector<float> > t_lin(n);
    std::vector<hhv::Profile> Tlin(n);
    for (int k = 0; k < n; ++k) {
      t_lin[k].resize((size_t)(db[k].L + 1) * 7);
      hhv::LinearTransitions(db[k].tr.data(), db[k].L, false, t_lin[k].data());
      Tlin[k] = T[k];
      Tlin[k].tr = t_lin[k].data();
    }
    std::vector<hhv::MacInput> in;
    std::vector<size_t> which;
    for (size_t h = 0; h < hits.size(); ++h) {
      if (hits[h].score < 25.f || hits[h].nsteps < 1) continue;
      hhv::MacInput m;
      m.entry = hits[h].entry;
      m.irep = hits[h].irep;
      m.i1 = hits[h].i1;
      m.j1 = hits[h].j1;
      m.i2 = hits[h].i2;
      m.j2 = hits[h].j2;
      m.nsteps = hits[h].nsteps;
      m.i = hits[h].i.data();
      m.j = hits[h].j.data();
      in.push_back(m);
      which.push_back(h);
    }
    hhv_params hp = {0, par.loc, par.egq, par.egt, par.shift, par.corr, par.ssw, par.ssm};
    hhv_ctx* ctx = nullptr;
    if (hhv_create(&ctx, &hp) != HHV_OK) throw hhv::Error(-1, hhv_last_error());
    hhv::MacParameters mp;
    mp.loc = par.loc;
    mp.shift = par.shift;
    std::vector<hhv::MacAlignment> mac;
    if (!in.empty()) mac = hhv::PosteriorDecoderRunner(ctx).executeComputation(mp, Qlin, Tlin, in);
    hhv_destroy(ctx);

    printf("%zu Viterbi hits (%d templates, altali %d), %zu realigned\n", hits.size(), n, par.altali, mac.size());
    for (size_t e = 0; e < mac.size(); ++e) {
      const hhv::Hit& v = hits[which[e]];
      printf("template %3d rep %d  viterbi score %7.2f  q %3d-%-3d t %3d-%-3d | MAC q %3d-%-3d t %3d-%-3d cols %3d sum_of_probs %6.2f\n",
             v.entry, v.irep, v.score, v.i1, v.i2, v.j1, v.j2, mac[e].i1, mac[e].i2, mac[e].j1, mac[e].j2, mac[e].matched_cols,
             mac[e].sum_of_probs);
    }
    int related_found = 0;
    for (size_t e = 0; e < mac.size(); ++e) related_found += (hits[which[e]].entry % 4 == 0 && hits[which[e]].irep == 1);
    if (related_found < n / 4 - 1) {
      fprintf(stderr, "only %d of %d related templates found\n", related_found, n / 4);
      return 2;
    }
  } catch (const hhv::Error& e) {
    fprintf(stderr, "hhv error %d: %s\n", e.status, e.what());
    return 1;
  }
  return 0;
}
