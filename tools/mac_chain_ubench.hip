// tools/mac_chain_ubench.hip -- what a step of the MAC kernels' serial chains costs a lone wavefront (round 5).
// The forward / backward kernels evaluate first-order recurrences along a row in the reference's order; this measures the
// candidates in isolation, one wave per SIMD (256 CUs x 4 workgroups of one wave), clocks per step = time x clock / steps:
//   0  IM sweep      y = c + shr1(y) * q * b          (2 DPP moves + 3 fp64)
//   1  GD sweep      y = a + shr1(y) * b              (2 DPP + 2 fp64)
//   2  total sweep   acc = shr1(acc) + f              (2 DPP + 1 fp64)
//   3  all three in one loop (the single-wave kernels' sweep)
//   4  two IM sweeps of independent rows in one loop
//   5  register walk  y = c + (y * q) * b, operands in registers (the dependent-latency floor of the IM chain)
//   6  LDS walk       the same with c, b read from LDS (ds_read2_b64 per two columns) and y stored
//   7  total walk     acc = acc + f, f from LDS
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-result -o build/mac_chain_ubench tools/mac_chain_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double shr1_dz(double y) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(y), 0x138, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(y), 0x138, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int KIND>
__global__ void __launch_bounds__(64) chain_kernel(const double* in, double* out, int reps) {
  __shared__ double lds[3 * 72];
  const int lane = threadIdx.x;
  double c = in[lane], b = in[64 + lane], q = in[128], f = in[192 + lane];
  double c2 = in[256 + lane], b2 = in[320 + lane];
  lds[lane] = c, lds[72 + lane] = b, lds[144 + lane] = f;
  __syncthreads();
  double y = 0.0, g = 0.0, acc = 1.0, y2 = 0.0;
  for (int r = 0; r < reps; ++r) {
    if (KIND <= 4) {
#pragma unroll 1
      for (int s = 0; s < 64; ++s) {
        if (KIND == 0 || KIND == 3 || KIND == 4) y = c + shr1_dz(y) * q * b;
        if (KIND == 1 || KIND == 3) g = c2 + shr1_dz(g) * b2;
        if (KIND == 2 || KIND == 3) acc = shr1_dz(acc) + f;
        if (KIND == 4) y2 = c2 + shr1_dz(y2) * q * b2;
      }
    } else if (KIND == 5) {
#pragma unroll 8
      for (int s = 0; s < 64; ++s) {
        double t = y * q;
        t = t * b;
        y = c + t;
      }
    } else if (KIND == 6) {
      volatile double* pc = lds;
      volatile double* pb = lds + 72;
#pragma unroll 8
      for (int s = 0; s < 64; ++s) {
        double t = y * q;
        t = t * pb[s];
        y = pc[s] + t;
        pc[s] = y;
      }
    } else {
      volatile double* pf = lds + 144;
#pragma unroll 8
      for (int s = 0; s < 64; ++s) acc = acc + pf[s];
    }
    // keep the chains bounded
    y *= 0.5, g *= 0.5, y2 *= 0.5;
    acc = acc * 0.25 + 1.0;
  }
  out[blockIdx.x * 64 + lane] = y + g + acc + y2;
}

template <int KIND>
static void run(const char* name, const double* d_in, double* d_out, double mhz) {
  const int reps = 2000, blocks = 1024;
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  chain_kernel<KIND><<<blocks, 64>>>(d_in, d_out, 10);
  hipEventRecord(e0);
  chain_kernel<KIND><<<blocks, 64>>>(d_in, d_out, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %.3f ms  %.1f clk per step (per column)\n", name, ms, ms * 1e-3 * mhz * 1e6 / (reps * 64.0));
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const double mhz = prop.clockRate / 1e3;
  printf("device %s  CUs %d  clock %.0f MHz; one wave per SIMD\n", prop.name, prop.multiProcessorCount, mhz);
  std::vector<double> h(512);
  for (int i = 0; i < 512; ++i) h[i] = 0.01 + 0.9 * ((i * 37) % 101) / 101.0;
  double *d_in, *d_out;
  hipMalloc(&d_in, 512 * 8), hipMalloc(&d_out, 1024 * 64 * 8);
  hipMemcpy(d_in, h.data(), 512 * 8, hipMemcpyHostToDevice);
  run<0>("IM sweep (2 DPP + mul mul add)", d_in, d_out, mhz);
  run<1>("GD sweep (2 DPP + mul add)", d_in, d_out, mhz);
  run<2>("total sweep (2 DPP + add)", d_in, d_out, mhz);
  run<3>("IM + GD + total in one loop", d_in, d_out, mhz);
  run<4>("two IM sweeps in one loop", d_in, d_out, mhz);
  run<5>("IM register walk (mul mul add)", d_in, d_out, mhz);
  run<6>("IM LDS walk", d_in, d_out, mhz);
  run<7>("total LDS walk (add)", d_in, d_out, mhz);
  return 0;
}
