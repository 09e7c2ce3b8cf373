// examples/synth8d.h -- the synthetic database of SURVEY.md 8(d) / BASELINE configs 2-4 in C++: the same definition as
// hh-suite_amd/pyhhv/synth_stream.py (which bench.py uses), so that the native programs time the same data.
//   template with GLOBAL id g: seed 0x5EED0000 + g -> splitmix64 (four outputs = the state) -> xoshiro256**;  query: seed 0x51000000
//   u = (x >> 40) * 2^-24;  per column 28 draws: 20 for the profile (g = erfinv(2u - 1 + 2^-24)^2 = Gamma(1/2) draws, normalised,
//   f = 0.7 g + 0.3 pb, p = f / pb for templates), 8 for the transitions (pI, pD ~ U[0.01, 0.05] from the first two)
//   M2I = 0.6 log2 pI, M2D = 0.6 log2 pD, M2M = log2(1 - pI - pD), I2M = D2M = log2 0.6, I2I = D2D = 0.6 log2 0.4; rows 0 and L
//   as AddTransitionPseudocounts leaves them.  (Bit-for-bit agreement with the torch version is not claimed: erfinv here is a
//   rational start + two Newton steps on erf, torch's is its own - the distributions, seeds and draw order are the same.)
// xoshiro256** / splitmix64: Blackman & Vigna, public domain.  Test / example code, not part of the product.
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

namespace synth8d {

static const double kPbRaw[20] = {0.0787, 0.0512, 0.0448, 0.0536, 0.0135, 0.0403, 0.0610, 0.0688, 0.0229, 0.0590,
                                  0.0964, 0.0593, 0.0237, 0.0396, 0.0483, 0.0683, 0.0585, 0.0132, 0.0321, 0.0668};

struct Xoshiro {
  uint64_t s[4];
  explicit Xoshiro(uint64_t seed) {
    uint64_t x = seed;
    for (int k = 0; k < 4; ++k) {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      s[k] = z ^ (z >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
  double u() { return (double)(next() >> 40) * (1.0 / 16777216.0); }
};

inline double erfinv(double x) {  // |x| < 1
  // Giles' single-precision approximation as the start, two Newton steps on erf (double): ~1e-15 relative
  double w = -log((1.0 - x) * (1.0 + x)), p;
  if (w < 5.0) {
    w -= 2.5;
    p = 2.81022636e-08;
    p = 3.43273939e-07 + p * w;
    p = -3.5233877e-06 + p * w;
    p = -4.39150654e-06 + p * w;
    p = 0.00021858087 + p * w;
    p = -0.00125372503 + p * w;
    p = -0.00417768164 + p * w;
    p = 0.246640727 + p * w;
    p = 1.50140941 + p * w;
  } else {
    w = sqrt(w) - 3.0;
    p = -0.000200214257;
    p = 0.000100950558 + p * w;
    p = 0.00134934322 + p * w;
    p = -0.00367342844 + p * w;
    p = 0.00573950773 + p * w;
    p = -0.0076224613 + p * w;
    p = 0.00943887047 + p * w;
    p = 1.00167406 + p * w;
    p = 2.83297682 + p * w;
  }
  double y = p * x;
  for (int it = 0; it < 2; ++it) y -= (erf(y) - x) / (1.1283791670955126 * exp(-y * y));
  return y;
}

struct Profile {
  int L = 0;
  std::vector<float> p, tr;  // p[(L+1)*20], tr[(L+1)*7] in the order of src/hhdecl.h:68 (M2M, M2I, M2D, I2M, I2I, D2M, D2D)
};

// is_query: column probabilities (no null model); else odds f / pb (what Viterbi::Align sees after PrepareTemplateHMM)
inline Profile make(uint64_t seed, int L, bool is_query) {
  double pb[20], sum = 0;
  for (int a = 0; a < 20; ++a) sum += kPbRaw[a];
  for (int a = 0; a < 20; ++a) pb[a] = (double)(float)(kPbRaw[a] / sum);
  Xoshiro rng(seed);
  Profile h;
  h.L = L;
  h.p.assign((size_t)(L + 1) * 20, 0.f);
  h.tr.assign((size_t)(L + 1) * 7, 0.f);
  std::vector<double> pI(L + 1, 0.01), pD(L + 1, 0.01);
  for (int i = 1; i <= L; ++i) {
    double g[20], gs = 0, f[20], fs = 0;
    for (int a = 0; a < 20; ++a) {
      const double e = erfinv(2.0 * rng.u() - 1.0 + 1.0 / 16777216.0);
      gs += (g[a] = e * e);
    }
    pI[i] = 0.01 + 0.04 * rng.u();
    pD[i] = 0.01 + 0.04 * rng.u();
    for (int d = 0; d < 6; ++d) (void)rng.next();  // (28 draws per column)
    for (int a = 0; a < 20; ++a) fs += (f[a] = 0.7 * g[a] / gs + 0.3 * pb[a]);
    for (int a = 0; a < 20; ++a) h.p[(size_t)i * 20 + a] = is_query ? (float)(f[a] / fs) : (float)(f[a] / fs / pb[a]);
  }
  for (int i = 0; i <= L; ++i) {
    float* t = &h.tr[(size_t)i * 7];
    t[0] = (float)log2(1.0 - pI[i] - pD[i]);
    t[1] = (float)(0.6 * log2(pI[i]));
    t[2] = (float)(0.6 * log2(pD[i]));
    t[3] = t[5] = (float)log2(0.6);
    t[4] = t[6] = (float)(0.6 * log2(0.4));
    if (i == 0 || i == L) {
      t[0] = 0.f;
      t[1] = t[2] = -100000.f;
    }
    if (i == L) {
      t[5] = 0.f;
      t[6] = -100000.f;
    }
  }
  return h;
}
inline Profile make_template(int64_t global_id, int L) { return make(0x5EED0000ull + (uint64_t)global_id, L, false); }
inline Profile make_query(int Lq) { return make(0x51000000ull, Lq, true); }

// BASELINE configs[4]: L = 49 + k, k ~ Zipf(1.2) truncated to 1..951, one draw per template from its own stream
inline int zipf_length(int64_t global_id) {
  static std::vector<double> cdf;
  if (cdf.empty()) {
    double s = 0;
    for (int k = 1; k <= 951; ++k) cdf.push_back(s += pow((double)k, -1.2));
    for (double& c : cdf) c /= s;
  }
  Xoshiro r(0x21F00000ull + (uint64_t)global_id);
  const double u = r.u();
  size_t k = 0;
  while (k < cdf.size() && cdf[k] <= u) ++k;
  return 49 + 1 + (int)(k < 951 ? k : 950);
}

}  // namespace synth8d
