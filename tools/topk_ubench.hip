// topk_ubench.hip -- the one-launch top-K of a small set (hhv_topk.hip: topk_small_kernel, merge_hits_small_kernel) on its own:
// kernel time by HIP events over many launches and, with -DHHV_TOPK_TIMING, the clock at its phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DHHV_TOPK_TIMING -Ihh-suite_amd/csrc -Iinclude tools/topk_ubench.hip -o build/topk_ubench
//   build/topk_ubench [n] [k]
#include "../hh-suite_amd/csrc/hhv_topk.hip"

#include <random>
#include <vector>

int main(int argc, char** argv) {
  using namespace hhv;
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
