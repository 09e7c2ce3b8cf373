// tools/lds_dma_probe.hip -- measurement aid, not product: does global_load_lds (LDS-DMA) reach LDS addresses beyond 64 KB on
// gfx950 (160 KB of LDS per CU; M0 carries the destination base)?  One workgroup, 150 KB of static LDS; for a list of byte
// offsets: clear the LDS, DMA 1 KiB of a pattern to that offset, read the whole LDS back and report where the pattern landed.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int WORDS = 150 * 1024 / 4;
__global__ void __launch_bounds__(64) probe(const float4* src, unsigned* out, int byte_off) {
  __shared__ unsigned lds[WORDS];
  for (int e = threadIdx.x; e < WORDS; e += 64) lds[e] = 0;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + threadIdx.x),
                                   (__attribute__((address_space(3))) void*)((char*)lds + byte_off), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < WORDS; e += 64) out[e] = lds[e];
}
int main() {
  std::vector<unsigned> pat(256);
  for (int i = 0; i < 256; ++i) pat[i] = 0xABC00000u + i;
  float4* d_src;
  unsigned* d_out;
  hipMalloc(&d_src, 1024);
  hipMalloc(&d_out, WORDS * 4);
  hipMemcpy(d_src, pat.data(), 1024, hipMemcpyHostToDevice);
  std::vector<unsigned> h(WORDS);
  for (int off : {0, 16384, 60 * 1024, 64 * 1024 - 512, 64 * 1024, 70 * 1024, 100352, 128 * 1024, 149 * 1024}) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, d_out, off);
    hipMemcpy(h.data(), d_out, WORDS * 4, hipMemcpyDeviceToHost);
    int first = -1, count = 0;
    for (int e = 0; e < WORDS; ++e)
      if (h[e] != 0) {
        if (first < 0) first = e;
        ++count;
      }
    printf("asked byte offset %6d: %d non-zero words, first at byte %d (%s)\n", off, count, first * 4,
           first * 4 == off && count == 256 && h[first] == 0xABC00000u ? "as asked" : "ELSEWHERE");
  }
  return 0;
}
