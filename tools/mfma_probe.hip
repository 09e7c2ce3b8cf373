// tools/mfma_probe.hip -- is a one-term MFMA product the IEEE fp32 product?  (measurement aid, not product)
// The Viterbi emission score needs the 20 products q[a] * t[a] rounded individually (the reference multiplies and adds
// separately).  An MFMA with a single non-zero term per output and C = 0 computes a * b + 0: if that equals fl(a * b)
// bit for bit - including products in the subnormal range - the products of a tile could come from the matrix pipe while
// the VALU only adds.  Checks v_mfma_f32_4x4x1_16b_f32 (K = 1) and v_mfma_f32_32x32x2_f32 with the second K slice zeroed
// against host products, as multisets per block (independent of the output layout).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k4x4(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[blockIdx.x * 64 + l], b[blockIdx.x * 64 + l], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[(blockIdx.x * 64 + l) * 4 + r] = c[r];
}
__global__ void k32(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f16v c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  const float av = l < 32 ? a[blockIdx.x * 32 + l] : 0.f, bv = l < 32 ? b[blockIdx.x * 32 + l] : 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) d[(blockIdx.x * 64 + l) * 16 + r] = c[r];
}

static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  const int NB = 4096;
  std::mt19937 rng(12345);
  std::vector<float> a(NB * 64), b(NB * 64);
  auto gen = [&](int mode) {
    std::uniform_real_distribution<float> u(0.f, 1.f);
    for (size_t k = 0; k < a.size(); ++k) {
      float x = u(rng), y = u(rng);
      if (mode == 1) { x = ldexpf(x, -(int)(rng() % 70)); y = ldexpf(y, -(int)(rng() % 70)); }   // products down to 2^-140
      if (mode == 2) { x = ldexpf(x, -60 - (int)(rng() % 10)); y = ldexpf(y, -60 - (int)(rng() % 10)); }  // subnormal products
      a[k] = x; b[k] = y;
    }
  };
  float *da, *db, *dd;
  (void)hipMalloc(&da, a.size() * 4); (void)hipMalloc(&db, b.size() * 4); (void)hipMalloc(&dd, (size_t)NB * 64 * 16 * 4);
  std::vector<float> d((size_t)NB * 64 * 16);
  for (int mode = 0; mode < 3; ++mode) {
    gen(mode);
    (void)hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    // 4x4x1: 16 blocks of 4 lanes per wave
    hipLaunchKernelGGL(k4x4, dim3(NB), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(d.data(), dd, (size_t)NB * 64 * 4 * 4, hipMemcpyDeviceToHost);
    long bad = 0, tot = 0, sub = 0;
    for (int w = 0; w < NB; ++w)
      for (int blk = 0; blk < 16; ++blk) {
        std::vector<uint32_t> ref, got;
        for (int i = 0; i < 4; ++i)
          for (int j = 0; j < 4; ++j) {
            volatile float p = a[w * 64 + blk * 4 + i] * b[w * 64 + blk * 4 + j];
            ref.push_back(bits(p));
            if (p != 0.f && fabsf(p) < 1.17549435e-38f) sub++;
          }
        for (int l = 0; l < 4; ++l)
          for (int r = 0; r < 4; ++r) got.push_back(bits(d[((size_t)w * 64 + blk * 4 + l) * 4 + r]));
        std::sort(ref.begin(), ref.end()); std::sort(got.begin(), got.end());
        tot += 16;
        for (int q = 0; q < 16; ++q) bad += ref[q] != got[q];
      }
    printf("mode %d  4x4x1   : %ld products, %ld subnormal on the host, %ld differ from fl(a*b)\n", mode, tot, sub, bad);
    // 32x32x2 with the second K slice zero
    hipLaunchKernelGGL(k32, dim3(NB), dim3(64), 0, 0, da, db, dd);
    (void)hipMemcpy(d.data(), dd, (size_t)NB * 64 * 16 * 4, hipMemcpyDeviceToHost);
    bad = tot = 0;
    for (int w = 0; w < NB; ++w) {
      std::vector<uint32_t> ref, got;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) { volatile float p = a[w * 32 + i] * b[w * 32 + j]; ref.push_back(bits(p)); }
      for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 16; ++r) got.push_back(bits(d[((size_t)w * 64 + l) * 16 + r]));
      std::sort(ref.begin(), ref.end()); std::sort(got.begin(), got.end());
      tot += 1024;
      for (int q = 0; q < 1024; ++q) bad += ref[q] != got[q];
    }
    printf("mode %d  32x32x2 : %ld products, %ld differ from fl(a*b)\n", mode, tot, bad);
  }
  return 0;
}
