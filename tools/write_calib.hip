// tools/write_calib.hip -- calibration of rocprofv3's WRITE_SIZE counter on gfx950 (measurement aid, not product).
// MI355X_MICROARCH.md (HBM section) calibrates FETCH_SIZE only and says WRITE_SIZE must be calibrated on a known byte
// count in one's own access pattern.  The backtrace variant of hhv_stream_kernel stores 8 bytes per lane per step:
// 512 contiguous bytes per wave per step, every wave walking its own range of the buffer.  The kernels below do
// exactly that (and the 16-bytes-per-lane alternative) for a known number of bytes:
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace -- tools/write_calib
// and tools/summarize_profile.py divides the counter by the bytes printed here.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(64) calib_store8(uint64_t* out, int steps, uint64_t v) {
  uint64_t* p = out + (size_t)blockIdx.x * steps * 64 + threadIdx.x;
  for (int s = 0; s < steps; ++s) p[(size_t)s * 64] = v + s;
}
__global__ void __launch_bounds__(64) calib_store16(ulonglong2* out, int steps, uint64_t v) {
  ulonglong2* p = out + (size_t)blockIdx.x * steps * 64 + threadIdx.x;
  for (int s = 0; s < steps; ++s) p[(size_t)s * 64] = make_ulonglong2(v + s, v);
}

int main() {
  const int waves = 2048, steps8 = 14700, steps16 = 7350;
  const size_t bytes = (size_t)waves * steps8 * 64 * 8;  // 15.4 GB, the size of the backtrace buffer of the 100 k x 300 benchmark
  void* buf = nullptr;
  if (hipMalloc(&buf, bytes) != hipSuccess) {
    printf("hipMalloc failed\n");
    return 1;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(calib_store8, dim3(waves), dim3(64), 0, 0, (uint64_t*)buf, steps8, (uint64_t)rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("calib_store8  bytes %zu  %.3f ms  %.1f GB/s\n", bytes, ms, bytes / (ms * 1e-3) / 1e9);
    hipEventRecord(e0);
    hipLaunchKernelGGL(calib_store16, dim3(waves), dim3(64), 0, 0, (ulonglong2*)buf, steps16, (uint64_t)rep);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("calib_store16 bytes %zu  %.3f ms  %.1f GB/s\n", bytes, ms, bytes / (ms * 1e-3) / 1e9);
  }
  hipFree(buf);
  return 0;
}
