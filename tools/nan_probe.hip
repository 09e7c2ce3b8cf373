// tools/nan_probe.hip -- what sign does the NaN of (-inf) - (-inf) carry on gfx950, and what do v_max_f32 / v_sub_f32 make of
// signed zeros?  (The sign-bit form of the backtrace flags, viterbi_lane.h BT_PAIR_SIGN, is exact as long as no operand is -0
// and the two operands are not both -inf; this probe records what the hardware does in those cases.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__global__ void k(const float* a, const float* b, uint32_t* out, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  float d, m;
  asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a[i]), "v"(b[i]));
  asm volatile("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(a[i]), "v"(b[i]));
  out[2 * i] = __float_as_uint(d);
  out[2 * i + 1] = __float_as_uint(m);
}
int main() {
  const float inf = INFINITY;
  const float ha[] = {-inf, -inf, 0.0f, -0.0f, 0.0f, -0.0f, -3.4028235e38f, 1.5f};
  const float hb[] = {-inf, inf, -0.0f, 0.0f, 0.0f, -0.0f, -3.4028235e38f, 1.5f};
  const int n = 8;
  float *a, *b;
  uint32_t* o;
  hipMalloc(&a, n * 4);
  hipMalloc(&b, n * 4);
  hipMalloc(&o, n * 8);
  hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, n);
  uint32_t ho[16];
  hipMemcpy(ho, o, n * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("a = %g  b = %g :  a - b = %08x   max(a, b) = %08x\n", ha[i], hb[i], ho[2 * i], ho[2 * i + 1]);
  return 0;
}
