// hhv_kernels_w32.hip -- instantiation unit of hhv_stream_kernel for short queries: the wavefront is 2 independent
// systolic arrays of 32 lanes (Lq <= 160), each walking its own range of the template stream (hhv_stream_kernel.h).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (like hhv_kernels.hip).
#include "hhv_stream_kernel.h"

namespace hhv {

void* stream_kernel_w32(int R, bool local, bool bt, bool celloff, bool ss) {
  return stream_kernel_pick<32>(R, local, bt, celloff, false, ss);
}

}  // namespace hhv
