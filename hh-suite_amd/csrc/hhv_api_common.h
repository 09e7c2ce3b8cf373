// hhv_api_common.h -- internals shared by the translation units of the C-ABI host layer (hhv_api*.cpp): the opaque
// handle structs, the thread-local error message and the small device-memory helpers.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hhviterbi_hip.h"
#include "hhv_internal.h"
#include "hhv_pack.h"
#include "viterbi_lane.h"

namespace hhv {
namespace api {

// sets the message hhv_last_error() returns (thread local) and hands the status code back
int fail(int code, const char* fmt, ...);
const char* last_error();

// hipStreamSynchronize on the context's stream, then the device error word: a kernel that met a failure (a pair wave that
// waited in vain for its partner, an illegal backtrace state) has set a bit there - HHV_E_DEVICE with text, the word is
// cleared.  Every entry point that hands device results to the host comes through here.
int sync_check(struct ::hhv_ctx* c, const char* who);
int check_error_word(struct ::hhv_ctx* c, const char* who);  // the device error word alone, without waiting for the stream (hhv_check_error)

template <typename T>
inline void dfree(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

// Device blocks of a context's template sets, kept between sets (round 6).  A search makes and frees a set per stage - the
// prepared subset, the gathered subset of a round, their results, backtrace planes and path pools: ~25 hipMalloc + hipFree
// each, and hipFree waits for the whole device - 4 ms of a 24 ms ViterbiRunner::alignment over 20 000 resident templates.
// tmalloc hands out a cached block of at least the size asked for (at most twice it) before it asks the runtime; tfree
// keeps blocks up to POOL_BLOCK_MAX, POOL_TOTAL_MAX in all, and records an event on the context's stream that the next owner
// waits for (the context's work is ordered by its one stream; a blocking copy into a recycled block is not).
struct DevPool {
  struct Block {
    void* p;
    hipEvent_t ev;
  };
  std::multimap<size_t, Block> free_blocks;     // by size
  std::unordered_map<void*, size_t> live;       // blocks handed out by pool_malloc
  std::vector<hipEvent_t> spare;
  size_t cached = 0;
  std::mutex m;
};
constexpr size_t POOL_BLOCK_MAX = (size_t)512 << 20, POOL_TOTAL_MAX = (size_t)4 << 30;
hipError_t pool_malloc(struct ::hhv_ctx* c, void** p, size_t bytes);
void pool_free(struct ::hhv_ctx* c, void* p);
bool context_alive(struct ::hhv_ctx* c);  // created and not yet destroyed (a set freed after its context frees its blocks directly)
void pool_release(struct ::hhv_ctx* c);  // hands every cached block back to the runtime (hhv_destroy; an allocation that failed)
template <typename T>
inline hipError_t tmalloc(struct ::hhv_ctx* c, T** p, size_t bytes) {
  return pool_malloc(c, reinterpret_cast<void**>(p), bytes);
}
template <typename T>
inline void tfree(struct ::hhv_ctx* c, T*& p) {
  if (p) pool_free(c, const_cast<void*>(static_cast<const void*>(p)));
  p = nullptr;
}

}  // namespace api
}  // namespace hhv

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return hhv::api::fail(e_ == hipErrorOutOfMemory ? HHV_E_MEMORY : HHV_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct hhv_ctx {
  hhv_params par;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  bool ev_records = true;   // HHV_EVENT_RECORDS=0: the DP's events attached to the launches (hipExtLaunchKernel) instead of recorded on the stream
  int num_cus = 0;
  // query: plan.P passes of 64 * plan.R(p) rows (one pass up to Lq = 320)
  int Lq = 0;
  hhv::StripPlan plan;
  float* d_qpack = nullptr;  // [plan.rows()][28]
  uint32_t* d_queue = nullptr;  // ticket counter of the stream kernel's work queue (set in front of each launch)
  float* d_qp = nullptr;     // [(Lq+1)][20] AoS, for the backtrace rescoring: behind the rows in the same device block
  size_t qpack_cap = 0;      // floats allocated behind d_qpack (kept between queries)
  void* q_stage = nullptr;           // pinned staging block of hhv_set_query
  size_t q_stage_bytes = 0;
  hipEvent_t ev_q = nullptr;         // the last query's copies have left the staging block
  bool q_stage_busy = false;
  bool q_dirty = false;              // the query is packed in the staging block, not yet on the device (flush_query, hhv_align_async)
  bool q_by_copy = false;            // HHV_QUERY_COPY=1: hipMemcpyAsync + ev_q in hhv_set_query instead of the upload kernel
  uint32_t q_seq = 0;                // number of upload kernels launched; the kernel writes it to h_err[1] when it has read the block
  size_t q_n4 = 0;                   // float4s of the staged query (rows + AoS profile)
  // fast_log2 tables (src/util-inl.h:108-130)
  float* d_lg2 = nullptr;
  float* d_diff = nullptr;
  // secondary structure
  std::vector<float> S73, S33, S37;                    // host copies of the score tables
  std::vector<int8_t> q_pred, q_conf, q_dssp;          // [Lq+1], empty = absent
  int ss_hmm_mode = 0;                                 // HMM::NO_SS_INFORMATION
  bool ss_dirty = true;
  // secondary-structure operands for the next MAC call (hhv_mac_set_ss), consumed by it
  bool mac_ss_pending = false;
  bool mac_lists = false;  // hhv_mac_set_lists: the following hhv_mac_realign* calls keep the forward / backward list planes
  int mac_ss_Lq = 0;
  std::vector<float> mac_ss_tab;
  std::vector<uint8_t> mac_ss_qidx, mac_ss_tidx;
  std::vector<int64_t> mac_ss_toff;
  std::vector<int32_t> mac_ss_mode;
  hhv::MacStreams mac_side = {};                          // side streams of the MAC length classes (created at the first use)
  bool mac_side_ready = false;
  hhv::api::DevPool pool;                                 // device blocks of freed template sets (tmalloc / tfree)
  // hhv_hit_paths_packed: device scratch and pinned mirror of the compact path records (grow-only, shared by the context's sets)
  void* d_packed = nullptr;
  size_t d_packed_bytes = 0;
  void* h_packed = nullptr;
  size_t h_packed_bytes = 0;
  void* mac_pinned = nullptr;                          // pinned staging buffer of the MAC inputs
  size_t mac_pinned_bytes = 0;
  void* mac_pinned_out = nullptr;                      // pinned buffer of the MAC paths coming back (recycled through hhv_macset_free)
  size_t mac_pinned_out_bytes = 0;
  void* mac_cache = nullptr;                           // one recycled device block of the MAC realignment
  size_t mac_cache_bytes = 0;
  float* d_ss_table = nullptr;                         // ssw * table of the current mode (SS_TABLE_MAX floats, allocated once)
  int32_t* d_ss_q_off = nullptr;                       // [plan.rows()], kept between queries (ss_q_cap entries)
  size_t ss_q_cap = 0;
  void* ss_stage = nullptr;                            // pinned staging block of ensure_ss: table + offsets travel with asynchronous copies
  size_t ss_stage_bytes = 0;
  hipEvent_t ev_ss = nullptr;                          // the last staging block has left the host
  bool ss_stage_busy = false;
  int ss_tab_n = 0;                                    // floats of the current table
  // hhv_set_celloff_paths: pinned staging + device scratch of the path arrays (grow-only), guarded by an event
  void* co_stage = nullptr;
  void* d_co = nullptr;
  size_t co_bytes = 0;
  hipEvent_t ev_co = nullptr;
  bool co_busy = false;
  int ss_t_shift = 0, ss_t_mask = 0;
  void* d_merge = nullptr;                             // hhv_merge_hits: merge_cap records + one int
  int merge_cap = 0;
  // device error word (hhv_internal.h DEV_ERR_*): one host-mapped dword the kernels store to when they meet a failure;
  // read by every call that has just waited for the stream (hhv::api::sync_check)
  uint32_t* h_err = nullptr;
  uint32_t* d_err = nullptr;                           // the device's address of *h_err
  // launch policy (hhv_set_launch_policy; the defaults may come from the environment, read ONCE in hhv_create: ADVICE r4)
  int pair_mode = -1;                                  // -1 the library chooses, 0 one launch per strip, 1 a pair launch wherever a pair kernel exists
  int pair_swap = 0;                                   // pair kernels: workgroups with this bit of their number set swap the strips of their waves; -1 none
  int blocks_per_cu = 0;                               // > 0: at most this many resident workgroups per CU (measurements)
  int trace_mode = -1;                                 // backtrace walk: -1 the library's choice (= 1 since round 5), 0 one lane per template, 1 one wavefront per template
};

struct hhv_tset {
  hhv_ctx* ctx = nullptr;
  int32_t n = 0;
  std::vector<int32_t> L;
  std::vector<int64_t> rec_off;  // [n+1]: header record of template k; rec_off[n] = terminal header
  int64_t n_records = 0;         // rec_off[n] + 1
  float* d_records = nullptr;
  bool owns_records = false;
  int64_t* d_rec_off = nullptr;
  int32_t* d_L = nullptr;
  hhv::DevResult* d_results = nullptr;
  // wave partition
  int n_waves = 0, n_range_slots = 0;  // non-empty stream ranges / ranges incl. the empty padding
  int64_t* d_wave_rec = nullptr;
  int64_t* d_seg = nullptr;      // [n_seg + 1][2] segment table of the work queue (hhv_stream_kernel.h DQ), built at the first launch
  int n_seg = 0;
  // backtrace bytes: [pass][record][lane] entries
  uint64_t* d_bt = nullptr;
  bool bt_valid = false;
  // the backtrace / cell-off variants overwrite the mask bytes with compare bits: once such a launch has run, every entry
  // of the buffer is stale as a MASK until it is cleared (hhv_set_celloff clears the whole buffer when this is set)
  bool bt_dirty = false;
  int bt_Lq = 0;
  int bt_mm = 0;  // encoding of the MM predecessor the last backtrace launch wrote (viterbi_lane.h bt_mm_mode)
  hhv::StripPlan bt_plan;
  // carry between the passes of a long query
  float4* d_carry = nullptr;
  float* d_carry_mi = nullptr;
  // trace outputs
  std::vector<int64_t> path_off;
  int path_Lq = -1;
  int64_t* d_path_off = nullptr;
  int32_t* d_i_steps = nullptr;
  int32_t* d_j_steps = nullptr;
  int8_t* d_states = nullptr;
  float* d_S = nullptr;
  float* d_Sss = nullptr;  // per-step secondary-structure scores (searches with secondary-structure information only)
  hhv::DevHit* d_hits = nullptr;
  bool hits_valid = false;
  // host copy of the whole path pool, fetched with five copies on the first hhv_hit_path after a trace (asking for
  // the paths of thousands of hits one by one would cost four small copies each)
  std::vector<int32_t> h_i_steps, h_j_steps;
  std::vector<int8_t> h_states;
  std::vector<float> h_S;
  std::vector<hhv::DevHit> h_hits;
  bool host_paths_valid = false;
  // hhv_hit_paths_packed: the paths alone, compact, in a pinned buffer of the context (valid until the next call for any set)
  std::vector<int64_t> pk_off;
  bool packed_valid = false;
  // top-k scratch
  hhv::DevHit* d_topk = nullptr;
  int topk_cap = 0;
  uint64_t* d_keys = nullptr;
  uint64_t* d_sorted = nullptr;
  void* d_sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  hhv::DevHit* d_raw_hits = nullptr;
  float* d_neff = nullptr;    // hhv_tset_set_neff: Neff_HMM of every template (HHV_TOPK_PVALUE)
  float* d_rank = nullptr;    // ... and the ranking keys of the last such hhv_topk
  float q_neff = 0.0f;
  int32_t* d_gids = nullptr;  // hhv_tset_set_global_ids: global template id of every entry (sharded databases)
};

struct hhv_rawset {
  hhv_ctx* ctx = nullptr;
  int32_t n = 0;
  std::vector<int32_t> L;
  std::vector<int64_t> rec_off;
  int64_t n_cols = 0;
  float* d_raw = nullptr;
  int64_t* d_raw_off = nullptr;  // [n+1] first column of every template in d_raw
  float* d_neff_hmm = nullptr;
  float* d_p_tmp = nullptr;
  float* d_tr_tmp = nullptr;
  float* d_pav = nullptr;
  float* d_pb = nullptr;
  float* d_R = nullptr;
  float* d_qpav = nullptr;
  // length classes of the prepare kernels: fused with small LDS (L <= 447), fused with large LDS (L <= 1300), split
  int32_t* d_ids[3] = {nullptr, nullptr, nullptr};
  int32_t n_ids[3] = {0, 0, 0};
  int32_t max_L[3] = {0, 0, 0};
  bool prepared = false;
  // pcm 2 with pcc != 1 (src/hhhmm.cpp:1903-1909): tau per raw column, evaluated on the host with libm's powf - the device has
  // no bit-exact powf - for the (pca, pcb, pcc) it was last asked for
  float* d_tau = nullptr;
  float tau_pca = 0, tau_pcb = 0, tau_pcc = 0;
};


namespace hhv {
namespace api {
// creates the host/device bookkeeping of a template set (record offsets, results) - hhv_api.cpp
int tset_init_common(hhv_ctx* c, hhv_tset* ts, int32_t n, const int32_t* L);
}  // namespace api
}  // namespace hhv
