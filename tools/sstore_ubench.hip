// tools/sstore_ubench.hip -- are scalar stores a usable side channel next to a VALU-bound loop on gfx950?  (measurement aid)
// Models the backtrace variant of hhv_stream_kernel: per step ~550 VALU instructions, 45 compares into SGPR pairs and 23
// s_store_dwordx4 (360 bytes per wave and step), three s_waitcnt lgkmcnt(0) per step (the LDS waits of the real kernel).
// Variants: 0 = VALU only, 1 = + v_cmp_e64 to SGPRs, 2 = + s_store_dwordx4, 3 = VALU + 45 x (v_cmp_e32 + v_addc) (today's way).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define VALU8 "v_add_f32 %0, %0, %9\n v_mul_f32 %1, %1, %9\n v_add_f32 %2, %2, %9\n v_mul_f32 %3, %3, %9\n v_add_f32 %4, %4, %9\n v_mul_f32 %5, %5, %9\n v_add_f32 %6, %6, %9\n v_mul_f32 %7, %7, %9\n"
#define CMP2 "v_cmp_gt_f32_e64 s[20:21], %0, %9\n v_cmp_gt_f32_e64 s[22:23], %1, %9\n"
#define ST "s_store_dwordx4 s[20:23], %10, 0x0\n"
#define ADDC2 "v_cmp_gt_f32_e32 vcc, %0, %9\n v_addc_co_u32_e32 %8, vcc, %8, %8, vcc\n v_cmp_gt_f32_e32 vcc, %1, %9\n v_addc_co_u32_e32 %8, vcc, %8, %8, vcc\n"

template <int KIND>
__global__ void __launch_bounds__(64, 2) k(float* out, uint64_t* masks, int steps, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b = 1.0f + threadIdx.x * 1e-7f;
  uint32_t acc = 0;
  uint64_t* row = masks + (size_t)blockIdx.x * steps * 48;  // wave-uniform
  for (int s = 0; s < steps; ++s) {
    uint64_t* p = row + (size_t)s * 48;
#pragma unroll
    for (int part = 0; part < 3; ++part) {
      // ~184 VALU per part (23 x 8) and a third of the compares / stores
#pragma unroll
      for (int u = 0; u < 23; ++u) {
        if (KIND == 0)
          asm volatile(VALU8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(acc) : "v"(b));
        if (KIND == 1 || (KIND == 2 && u >= 8))
          asm volatile(VALU8 CMP2 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(acc) : "v"(b) : "s20", "s21", "s22", "s23");
        if (KIND == 2 && u < 8)
          asm volatile(VALU8 CMP2 ST : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(acc) : "v"(b), "s"(p + part * 16 + u * 2) : "s20", "s21", "s22", "s23", "memory");
        if (KIND == 3)
          asm volatile(VALU8 ADDC2 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(acc) : "v"(b) : "vcc");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  if (KIND == 2) asm volatile("s_dcache_wb" ::: "memory");
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)acc;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int waves = prop.multiProcessorCount * 8, steps = 2000;
  float* out;
  uint64_t* masks;
  hipMalloc(&out, (size_t)waves * 64 * 4);
  hipMalloc(&masks, (size_t)waves * steps * 48 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[4] = {"VALU only (552 per step)", "+ 69 x 2 v_cmp_e64 -> SGPR", "+ 24 s_store_dwordx4 per step", "+ 69 x 2 (v_cmp_e32 + v_addc)"};
  for (int kind = 0; kind < 4; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, out, masks, steps, 1.0f);
      if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, out, masks, steps, 1.0f);
      if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, out, masks, steps, 1.0f);
      if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, out, masks, steps, 1.0f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%-34s %.3f ms  -> %.0f clk per step per wave pair at 2.4 GHz\n", names[kind], ms, ms * 1e-3 * 2.4e9 / steps);
    }
  }
  // spot check of what the scalar stores wrote (after s_dcache_wb)
  uint64_t h[4];
  hipMemcpy(h, masks + 5 * 48, sizeof(h), hipMemcpyDeviceToHost);
  printf("masks[5][0..3] = %016llx %016llx %016llx %016llx  (%s)\n", (unsigned long long)h[0], (unsigned long long)h[1], (unsigned long long)h[2],
         (unsigned long long)h[3], hipGetErrorString(hipGetLastError()));
  return 0;
}
