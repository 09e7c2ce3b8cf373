// tools/sweep_ubench.hip -- the first-order recurrences along a row of the MAC forward pass (y_j = a_j + y_{j-1} * b_j, three
// of them per strip of 64 columns), evaluated in the reference's sequential order by ONE wavefront per hit (measurement aid):
//   A  the DPP sweep of hhv_mac.hip: 64 lanes recompute from their left neighbour n times (6 DPP moves + 7 f64 ops per step)
//   B  coefficients through LDS, lanes 0..2 each walk one recurrence (3 ds_read_b64 + 3 f64 ops + 1 ds_write_b64 per step)
// 500 blocks of one wave (the occupancy of a 500-hit realignment), 300 rows x 5 strips x 64 steps.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ double shr1_dz(double y) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(y), 0x138, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(y), 0x138, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int KIND>
__global__ void __launch_bounds__(64) k(double* out, int rows, int strips, double seed) {
  __shared__ double co[5][64];
  __shared__ double res[3][64];
  const int lane = threadIdx.x;
  double a_gd = seed + lane * 1e-3, b_gd = 0.5 + lane * 1e-4, c_im = seed * 0.5 + lane * 2e-3, b_im = 0.25 + lane * 1e-4, f_mm = 1e-3 * lane;
  const double qI2I = 0.75;
  double tot = 0.0;
  for (int i = 0; i < rows; ++i) {
    for (int st = 0; st < strips; ++st) {
      double gd = 0.0, im = 0.0, acc = tot;
      if (KIND == 0) {
        for (int s = 0; s < 64; ++s) {
          const double gl = shr1_dz(gd), il = shr1_dz(im);
          gd = a_gd + gl * b_gd;
          im = c_im + il * qI2I * b_im;
          acc = shr1_dz(acc) + f_mm;
        }
      } else {
        co[0][lane] = a_gd;
        co[1][lane] = b_gd;
        co[2][lane] = c_im;
        co[3][lane] = b_im;
        co[4][lane] = f_mm;
        __syncthreads();
        if (lane < 3) {
          // lane 0: y = (y * b_gd) * 1 + a_gd; lane 1: y = (y * qI2I) * b_im + c_im; lane 2: y = (y * 1) * 1 + f_mm
          const double* A = lane == 0 ? co[0] : lane == 1 ? co[2] : co[4];
          const double* M1 = lane == 0 ? co[1] : nullptr;
          const double* M2 = lane == 1 ? co[3] : nullptr;
          const double m1c = lane == 1 ? qI2I : 1.0;
          double y = lane == 2 ? tot : 0.0;
#pragma unroll 8
          for (int s = 0; s < 64; ++s) {
            const double m1 = M1 ? M1[s] : m1c, m2 = M2 ? M2[s] : 1.0;
            y = (y * m1) * m2 + A[s];
            res[lane][s] = y;
          }
        }
        __syncthreads();
        gd = res[0][lane];
        im = res[1][lane];
        acc = res[2][lane];
      }
      tot = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(acc), 63), __builtin_amdgcn_readlane(__double2loint(acc), 63)) * 1e-3;
      a_gd = a_gd * 0.999 + gd * 1e-6;
      c_im = c_im * 0.999 + im * 1e-6;
    }
  }
  out[blockIdx.x * 64 + lane] = a_gd + c_im + tot;
}

int main() {
  double* out;
  hipMalloc(&out, 500 * 64 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(500), dim3(64), 0, 0, out, 300, 5, 1.0);
      else hipLaunchKernelGGL(k<1>, dim3(500), dim3(64), 0, 0, out, 300, 5, 1.0);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.3f ms -> %.1f clk per sweep step at 2.4 GHz\n", kind ? "B in-lane via LDS" : "A DPP sweep       ", ms, ms * 1e-3 * 2.4e9 / (300.0 * 5 * 64));
    }
  return 0;
}
