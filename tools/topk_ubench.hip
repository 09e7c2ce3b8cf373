// topk_ubench.hip -- the one-launch top-K of a small set (hhv_topk.hip: topk_small_kernel, merge_hits_small_kernel) on its own:
// kernel time by HIP events over many launches and, with -DHHV_TOPK_TIMING, the clock at its phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHHV_TOPK_TIMING -Ihh-suite_amd/csrc -Iinclude tools/topk_ubench.hip -o build/topk_ubench
//   build/topk_ubench [n] [k]
#include "../hh-suite_amd/csrc/hhv_topk.hip"

#include <random>
#include <vector>


// ---- fuzz mode: `topk_ubench fuzz <seconds> [seed]` - random sets through both kernels (both branches of the top-K, record and hit
// sources, ranking keys, global ids; the merge with padding records) against std::sort on the host.  Prints one JSON line; exit code 1
// on the first mismatch (tests/test_gpu_topk_fuzz.py).
#include <algorithm>
#include <chrono>
#include <cstring>
static uint64_t host_key(float score, uint32_t idx) {
  uint32_t u;
  memcpy(&u, &score, 4);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - idx);
}
static int fuzz(double seconds, unsigned seed) {
  using namespace hhv;
  std::mt19937 rng(seed);
  const int NMAX = SEL_CHUNK;
  DevResult* d_res;
  DevHit *d_hits, *d_out, *d_in, *d_out2;
  float* d_rank;
  int32_t* d_gids;
  int* d_n;
  (void)hipMalloc(&d_res, NMAX * sizeof(DevResult));
  (void)hipMalloc(&d_hits, NMAX * sizeof(DevHit));
  (void)hipMalloc(&d_rank, NMAX * sizeof(float));
  (void)hipMalloc(&d_gids, NMAX * sizeof(int32_t));
  (void)hipMalloc(&d_out, 8192 * sizeof(DevHit));
  (void)hipMalloc(&d_in, 8192 * sizeof(DevHit));
  (void)hipMalloc(&d_out2, 8192 * sizeof(DevHit));
  (void)hipMalloc(&d_n, 4);
  const int edge_n[] = {1, 2, 63, 64, 65, 127, 128, 129, 1023, 1024, 1025, 2047, 2048, 4096, 4097, 10000, 16383, 16384};
  long cases = 0, radix_cases = 0, merge_cases = 0;
  const auto t0 = std::chrono::steady_clock::now();
  auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
  std::vector<float> score(NMAX), rank(NMAX);
  std::vector<DevResult> res(NMAX);
  std::vector<DevHit> hits(NMAX), out(8192);
  std::vector<int32_t> gids(NMAX);
  while (elapsed() < seconds) {
    // ---- top-K of a small set
    const int n = (rng() % 3 == 0) ? edge_n[rng() % (sizeof(edge_n) / sizeof(int))] : 1 + (int)(rng() % NMAX);
    int k = 1 + (int)(rng() % std::min(n, (int)SEL_KMAX));
    if (rng() % 4 == 0) k = std::min(n, (int)SEL_KMAX);
    if (rng() % 8 == 0) k = std::min(n, 500);
    const int pattern = (int)(rng() % 8), src = (int)(rng() % 3) /* 0 results, 1 hits, 2 hits + rank */, force = (int)(rng() % 3 == 0);
    const bool with_gids = rng() % 2;
    std::normal_distribution<float> nd(40.0f, 25.0f);
    const int distinct = 1 + (int)(rng() % 40);
    for (int i = 0; i < n; ++i) {
      float v;
      switch (pattern) {
        case 0: v = nd(rng); break;
        case 1: v = (float)(rng() % distinct) * 1.25f - 7.0f; break;       // few distinct scores: the index decides
        case 2: v = 17.25f; break;                                             // all equal
        case 3: v = (float)i * 0.01f; break;                                   // ascending: the best keys in the last threads
        case 4: v = (float)(n - i) * 0.01f; break;                             // descending
        case 5: v = (i % 1024 < 8) ? 100.0f + nd(rng) : nd(rng); break;         // the best keys crowd a few threads
        case 6: v = (rng() % 16 == 0) ? -FLT_MAX : (rng() % 16 == 1 ? FLT_MAX : nd(rng)); break;
        default: v = (i / 16 == (int)(rng() % 4)) ? 1e-30f * nd(rng) : -nd(rng) * nd(rng); break;
      }
      score[i] = v;
      rank[i] = (src == 2) ? -v + (float)(rng() % 3) : 0.0f;   // another order than the score's
      res[i] = DevResult{v, (int32_t)(rng() % 300), (int32_t)(rng() % 300), i};
      hits[i].score = v;
      hits[i].viterbi_score = v + 1;
      hits[i].score_ss = 0.5f;
      hits[i].index = i;
      hits[i].i1 = i;
      hits[i].j1 = 2 * i;
      hits[i].i2 = res[i].i2;
      hits[i].j2 = res[i].j2;
      hits[i].nsteps = 7;
      hits[i].matched_cols = 5;
      gids[i] = (int32_t)(3 * (n - i) + 11);
    }
    (void)hipMemcpy(d_res, res.data(), n * sizeof(DevResult), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_hits, hits.data(), n * sizeof(DevHit), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_rank, rank.data(), n * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_gids, gids.data(), n * sizeof(int32_t), hipMemcpyHostToDevice);
    (void)hipMemset(d_out, 0xEE, k * sizeof(DevHit));
    if (src == 0)
      hipLaunchKernelGGL(topk_small_kernel<SRC_RESULTS>, dim3(1), dim3(SEL_THREADS), 0, 0, (const void*)d_res, (const float*)nullptr, n, k, with_gids ? d_gids : nullptr, d_out, force);
    else
      hipLaunchKernelGGL(topk_small_kernel<SRC_HITS>, dim3(1), dim3(SEL_THREADS), 0, 0, (const void*)d_hits, src == 2 ? d_rank : nullptr, n, k, with_gids ? d_gids : nullptr, d_out, force);
    if (hipMemcpy(out.data(), d_out, k * sizeof(DevHit), hipMemcpyDeviceToHost) != hipSuccess) {
      printf("{\"error\": \"%s\"}\n", hipGetErrorString(hipGetLastError()));
      return 1;
    }
    std::vector<uint64_t> keys(n);
    for (int i = 0; i < n; ++i) keys[i] = host_key(src == 2 ? rank[i] : score[i], (uint32_t)i);
    std::sort(keys.begin(), keys.end(), std::greater<uint64_t>());
    for (int t = 0; t < k; ++t) {
      const int idx = (int)(0xFFFFFFFFu - (uint32_t)(keys[t] & 0xFFFFFFFFu));
      const int want_index = with_gids ? gids[idx] : idx;
      const bool ok = out[t].index == want_index && memcmp(&out[t].score, &score[idx], 4) == 0 && out[t].i2 == res[idx].i2 && out[t].j2 == res[idx].j2 &&
                      (src == 0 ? (out[t].nsteps == 0 && out[t].i1 == 0) : (out[t].i1 == idx && out[t].nsteps == 7));
      if (!ok) {
        printf("{\"mismatch\": \"topk\", \"n\": %d, \"k\": %d, \"pattern\": %d, \"src\": %d, \"force_radix\": %d, \"rank\": %d, \"got_index\": %d, \"want_index\": %d, \"cases\": %ld}\n",
               n, k, pattern, src, force, t, out[t].index, want_index, cases);
        return 1;
      }
    }
    ++cases;
    radix_cases += force;
    // ---- the last level of a large set: keys a selection level wrote (with padding zeros), records gathered from the source
    {
      const int nk = std::min(n, 1 + (int)(rng() % NMAX));               // real keys
      const int m2 = std::min(NMAX, nk + (int)(rng() % (1 + nk / 2)));    // slots, the rest padding
      std::vector<int> pick(n);
      for (int i = 0; i < n; ++i) pick[i] = i;
      std::shuffle(pick.begin(), pick.end(), rng);
      std::vector<uint64_t> kin(m2, 0);
      std::vector<int> slots(m2);
      for (int i = 0; i < m2; ++i) slots[i] = i;
      std::shuffle(slots.begin(), slots.end(), rng);
      for (int i = 0; i < nk; ++i) kin[slots[i]] = host_key(score[pick[i]], (uint32_t)pick[i]);
      const int k2 = 1 + (int)(rng() % std::min(nk, (int)SEL_KMAX));
      static uint64_t* d_kin = nullptr;
      if (!d_kin) (void)hipMalloc(&d_kin, NMAX * sizeof(uint64_t));
      (void)hipMemcpy(d_kin, kin.data(), m2 * sizeof(uint64_t), hipMemcpyHostToDevice);
      (void)hipMemset(d_out, 0xEE, k2 * sizeof(DevHit));
      const bool from_res = rng() % 2;
      if (from_res)
        hipLaunchKernelGGL((topk_small_kernel<SRC_RESULTS, true>), dim3(1), dim3(SEL_THREADS), 0, 0, (const void*)d_res, (const float*)nullptr, m2, k2, (const int32_t*)nullptr, d_out, force, (const uint64_t*)d_kin);
      else
        hipLaunchKernelGGL((topk_small_kernel<SRC_HITS, true>), dim3(1), dim3(SEL_THREADS), 0, 0, (const void*)d_hits, (const float*)nullptr, m2, k2, (const int32_t*)nullptr, d_out, force, (const uint64_t*)d_kin);
      (void)hipMemcpy(out.data(), d_out, k2 * sizeof(DevHit), hipMemcpyDeviceToHost);
      std::sort(kin.begin(), kin.end(), std::greater<uint64_t>());
      for (int t = 0; t < k2; ++t) {
        const int idx = (int)(0xFFFFFFFFu - (uint32_t)(kin[t] & 0xFFFFFFFFu));
        if (out[t].index != idx || memcmp(&out[t].score, &score[idx], 4) != 0 || out[t].i2 != res[idx].i2) {
          printf("{\"mismatch\": \"topk keys-in\", \"slots\": %d, \"keys\": %d, \"k\": %d, \"force_radix\": %d, \"rank\": %d, \"got_index\": %d, \"want_index\": %d, \"cases\": %ld}\n",
                 m2, nk, k2, force, t, out[t].index, idx, cases);
          return 1;
        }
      }
    }
    // ---- merge of m records (some of them padding), k of them wanted
    {
      const int m = (rng() % 2) ? 1 + (int)(rng() % 1024) : 1 + (int)(rng() % 4096);
      const int km = 1 + (int)(rng() % (rng() % 4 == 0 ? 2 * m : m));
      std::vector<DevHit> in(m);
      std::vector<int> perm(4 * m);
      for (int i = 0; i < 4 * m; ++i) perm[i] = i;
      std::shuffle(perm.begin(), perm.end(), rng);
      const int padmod = 2 + (int)(rng() % 20);
      for (int i = 0; i < m; ++i) {
        in[i] = hits[i % n];
        in[i].score = (pattern == 2) ? 17.25f : (rng() % 3 == 0 ? 17.25f : nd(rng));
        in[i].index = perm[i];
        in[i].i1 = i;
        if (rng() % padmod == 0) memset(&in[i], 0xFF, sizeof(DevHit));
      }
      (void)hipMemcpy(d_in, in.data(), m * sizeof(DevHit), hipMemcpyHostToDevice);
      (void)hipMemset(d_out2, 0xEE, std::min(km, 8192) * sizeof(DevHit));
      const int kk = std::min(km, 8192);
      if (m <= SEL_THREADS)
        hipLaunchKernelGGL(merge_hits_small_kernel, dim3(1), dim3(SEL_THREADS), 0, 0, (const DevHit*)d_in, m, kk, d_out2, d_n);
      else
        hipLaunchKernelGGL(merge_hits_kernel, dim3(1), dim3(1024), 0, 0, (const DevHit*)d_in, m, kk, d_out2, d_n);
      int nv = -1;
      (void)hipMemcpy(out.data(), d_out2, kk * sizeof(DevHit), hipMemcpyDeviceToHost);
      (void)hipMemcpy(&nv, d_n, 4, hipMemcpyDeviceToHost);
      std::vector<std::pair<uint64_t, int>> v;
      for (int i = 0; i < m; ++i)
        if (in[i].index >= 0) v.push_back({host_key(in[i].score, (uint32_t)in[i].index), i});
      std::sort(v.begin(), v.end(), [](const std::pair<uint64_t, int>& a, const std::pair<uint64_t, int>& b) { return a.first > b.first; });
      const int want_nv = std::min((int)v.size(), kk);
      bool ok = nv == want_nv;
      for (int t = 0; ok && t < kk; ++t) {
        if (t < want_nv) ok = memcmp(&out[t], &in[v[t].second], sizeof(DevHit)) == 0;
        else ok = out[t].index == -1 && out[t].nsteps == -1;
      }
      if (!ok) {
        printf("{\"mismatch\": \"merge\", \"m\": %d, \"k\": %d, \"n_valid\": %d, \"want_valid\": %d, \"cases\": %ld}\n", m, kk, nv, want_nv, merge_cases);
        return 1;
      }
      ++merge_cases;
    }
  }
  printf("{\"topk_small_cases\": %ld, \"of_them_radix_branch_forced\": %ld, \"merge_cases\": %ld, \"mismatches\": 0, \"seconds\": %.1f, \"seed\": %u}\n", cases, radix_cases,
         merge_cases, elapsed(), seed);
  return 0;
}

int main(int argc, char** argv) {
  using namespace hhv;
  if (argc > 1 && strcmp(argv[1], "fuzz") == 0) return fuzz(argc > 2 ? atof(argv[2]) : 10.0, argc > 3 ? (unsigned)atoi(argv[3]) : 12345u);
  const int n = argc > 1 ? atoi(argv[1]) : 10000, k = argc > 2 ? atoi(argv[2]) : 500;
  std::mt19937 rng(7);
  std::normal_distribution<float> nd(40.0f, 25.0f);
  std::vector<DevResult> res(n);
  for (int i = 0; i < n; ++i) res[i] = DevResult{nd(rng), 300, 300, i};
  DevResult* d_res;
  DevHit *d_out, *d_out2;
  int* d_n;
  hipMalloc(&d_res, n * sizeof(DevResult));
  hipMalloc(&d_out, 1024 * sizeof(DevHit));
  hipMalloc(&d_out2, 1024 * sizeof(DevHit));
  hipMalloc(&d_n, 4);
  hipMemcpy(d_res, res.data(), n * sizeof(DevResult), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int force = 0; force < 2; ++force) {
    const int reps = 200;
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(topk_small_kernel<SRC_RESULTS>, dim3(1), dim3(SEL_THREADS), 0, 0, (const void*)d_res, (const float*)nullptr, n, k, (const int32_t*)nullptr, d_out, force);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(topk_small_kernel<SRC_RESULTS>, dim3(1), dim3(SEL_THREADS), 0, 0, (const void*)d_res, (const float*)nullptr, n, k, (const int32_t*)nullptr, d_out, force);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("topk_small n %d k %d %s: %.2f us per launch (back to back)\n", n, k, force ? "radix branch" : "bound branch", ms * 1e3 / reps);
#ifdef HHV_TOPK_TIMING
    unsigned long long clk[16];
    hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_topk_clk), sizeof(clk));
    const char* names[6] = {"load keys", "sort of the thread maxima", "bound, count, (radix)", "compaction", "sort of the candidates", "gather"};
    for (int s = 0; s < 6; ++s) printf("   %-28s %8llu clk\n", names[s], clk[s + 1] - clk[s]);
    printf("   candidates %llu, total %llu clk (s_memtime: 100 MHz units)\n", clk[15], clk[6] - clk[0]);
#endif
  }
  {
    const int reps = 200;
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(merge_hits_small_kernel, dim3(1), dim3(SEL_THREADS), 0, 0, (const DevHit*)d_out, k, k, d_out2, d_n);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(merge_hits_small_kernel, dim3(1), dim3(SEL_THREADS), 0, 0, (const DevHit*)d_out, k, k, d_out2, d_n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("merge_hits_small m %d: %.2f us per launch\n", k, ms * 1e3 / reps);
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(merge_hits_kernel, dim3(1), dim3(1024), 0, 0, (const DevHit*)d_out, k, k, d_out2, d_n);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("merge_hits_kernel (LDS network) m %d: %.2f us per launch\n", k, ms * 1e3 / reps);
  }
  return 0;
}
