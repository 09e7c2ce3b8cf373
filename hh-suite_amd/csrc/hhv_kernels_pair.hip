// hhv_kernels_pair.hip -- instantiation unit of hhv_pair_kernel: two neighbouring strips of a query aligned in ONE launch by
// workgroups of two wavefronts, a 128-lane systolic array (hhv_stream_kernel.h: PairLds, hhv_pair_kernel) - the whole query
// (321 .. 640 rows), or one link of the chain of launches of a longer one.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize (like hhv_kernels.hip).
#include <atomic>

#include "hhv_stream_kernel.h"

namespace hhv {

// the ...AndSS pair kernels (hhv_ss_pair_kernel: four pairs per workgroup around one LDS copy of the score table): strips of
// three and four rows per lane (five-row strips have no registers left for the table values next to the FIFO row)
template <bool LOCAL, bool BT, int CHAIN>
static void* ss_pair_kernel_ptr(int R0, int R1) {
  if (R0 == 3 && R1 == 3) return (void*)hhv_ss_pair_kernel<3, 3, LOCAL, BT, CHAIN>;
  if (R0 == 4 && R1 == 3) return (void*)hhv_ss_pair_kernel<4, 3, LOCAL, BT, CHAIN>;
  if (R0 == 4 && R1 == 4) return (void*)hhv_ss_pair_kernel<4, 4, LOCAL, BT, CHAIN>;
  return nullptr;
}
template <bool LOCAL, bool BT, int CHAIN>
static void* pair_kernel_ptr(int R0, int R1) {
  if (R0 == 3 && R1 == 3) return (void*)hhv_pair_kernel<3, 3, LOCAL, BT, CHAIN>;
  if (R0 == 4 && R1 == 3) return (void*)hhv_pair_kernel<4, 3, LOCAL, BT, CHAIN>;
  if (R0 == 4 && R1 == 4) return (void*)hhv_pair_kernel<4, 4, LOCAL, BT, CHAIN>;
  // (five rows per lane with backtrace park query rows in LDS: those strips stay launches of their own; so do the five-row
  // strips of local-mode chains - a link's extra carry path does not fit into the 256 registers next to the per-row best)
  if (!BT && !(LOCAL && CHAIN != 0)) {
    if (R0 == 5 && R1 == 4) return (void*)hhv_pair_kernel<5, 4, LOCAL, false, (LOCAL ? 0 : CHAIN)>;
    if (R0 == 5 && R1 == 5) return (void*)hhv_pair_kernel<5, 5, LOCAL, false, (LOCAL ? 0 : CHAIN)>;
  }
  return nullptr;
}
template <int CHAIN>
static void* pair_kernel_pick(int R0, int R1, bool local, bool bt, bool ss) {
  if (ss) {
    if (bt) return local ? ss_pair_kernel_ptr<true, true, CHAIN>(R0, R1) : ss_pair_kernel_ptr<false, true, CHAIN>(R0, R1);
    return local ? ss_pair_kernel_ptr<true, false, CHAIN>(R0, R1) : ss_pair_kernel_ptr<false, false, CHAIN>(R0, R1);
  }
  if (bt) return local ? pair_kernel_ptr<true, true, CHAIN>(R0, R1) : pair_kernel_ptr<false, true, CHAIN>(R0, R1);
  return local ? pair_kernel_ptr<true, false, CHAIN>(R0, R1) : pair_kernel_ptr<false, false, CHAIN>(R0, R1);
}

// chain: bit 0 = not the first link of a chain of launches, bit 1 = not the last (hhv_pair_kernel)
void* pair_kernel(int R0, int R1, bool local, bool bt, int chain, bool ss) {
  switch (chain & 3) {
    case 0: return pair_kernel_pick<0>(R0, R1, local, bt, ss);
    case 1: return pair_kernel_pick<1>(R0, R1, local, bt, ss);
    case 2: return pair_kernel_pick<2>(R0, R1, local, bt, ss);
    default: return pair_kernel_pick<3>(R0, R1, local, bt, ss);
  }
}

// n_pairs: two-wave systolic arrays to start (ss: four of them per workgroup - hhv_ss_pair_kernel; the caller rounds up)
int launch_pair(int R0, int R1, bool local, bool bt, int chain, bool ss, const StreamArgs& a, int n_pairs, void* stream, void* ev_start,
                void* ev_stop) {
  void* fn = pair_kernel(R0, R1, local, bt, chain, ss);
  if (!fn) return -1;
  StreamArgs args = a;
  void* kargs[] = {&args};
  const dim3 grid(ss ? (n_pairs + SS_PAIRS - 1) / SS_PAIRS : n_pairs), block(ss ? SS_WAVES * LANES : 2 * LANES);
  // (events on the kernel's own dispatch: launch_stream)
  const hipError_t e = (ev_start || ev_stop) ? hipExtLaunchKernel(fn, grid, block, kargs, 0, (hipStream_t)stream, (hipEvent_t)ev_start, (hipEvent_t)ev_stop, 0)
                                             : hipLaunchKernel(fn, grid, block, kargs, 0, (hipStream_t)stream);
  return e == hipSuccess ? 0 : -(int)e;
}
int pair_kernel_pairs_per_workgroup(bool ss) { return ss ? SS_PAIRS : 1; }

// PAIRS (two-wave systolic arrays) a CU holds; 0 = no pair kernel for these strips
int pair_kernel_occupancy(int R0, int R1, bool local, bool bt, int chain, bool ss) {
  void* fn = pair_kernel(R0, R1, local, bt, chain, ss);
  if (!fn) return 0;
  // asked once per kernel (every search of a multi-strip query comes through here; the devices of a process are of one kind).
  // Concurrent first calls store the same value.
  static std::atomic<int> cache[6][6][2][2][4][2];
  std::atomic<int>& slot = cache[R0][R1][local ? 1 : 0][bt ? 1 : 0][chain & 3][ss ? 1 : 0];
  int nb = slot.load(std::memory_order_relaxed) - 1;  // (0 = not asked yet)
  if (nb >= 0) return nb;
  nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, ss ? SS_WAVES * LANES : 2 * LANES, 0) != hipSuccess) return 0;
  nb *= ss ? SS_PAIRS : 1;
  slot.store(nb + 1, std::memory_order_relaxed);
  return nb;
}

}  // namespace hhv
