// hhv_api.cpp -- C-ABI host layer of libhhviterbi_hip.so (declared in include/hhviterbi_hip.h): contexts, template sets,
// the wave partition of the template stream, the Viterbi launches, backtrace / hits / top-K.  The widened rows live in
// hhv_api_db.cpp (N1), hhv_api_prep.cpp (N2), hhv_api_prefilter.cpp (N3), hhv_api_mac.cpp (N4).
// There is deliberately no CPU compute path here: every entry point that needs arithmetic launches a HIP kernel and
// fails with HHV_E_DEVICE when no device is usable.
#include "hhv_api_common.h"
#include <stdlib.h>

#include <atomic>
#include <thread>

using namespace hhv;
using hhv::api::dfree;
using hhv::api::tfree;
using hhv::api::tmalloc;
using hhv::api::fail;
using hhv::api::sync_check;

// contexts that exist: a set freed after its context (against the header's rule) must not touch the context's pool
namespace {
std::mutex g_live_m;
std::vector<hhv_ctx*> g_live;
bool ctx_alive(hhv_ctx* c) {
  std::lock_guard<std::mutex> lock(g_live_m);
  return std::find(g_live.begin(), g_live.end(), c) != g_live.end();
}
}  // namespace
bool hhv::api::context_alive(hhv_ctx* c) { return c && ctx_alive(c); }

using hhv::api::tset_init_common;

static_assert(sizeof(hhv_result) == sizeof(DevResult), "hhv_result layout");
static_assert(sizeof(hhv_hit) == sizeof(DevHit), "hhv_hit layout");
static_assert(HHV_STREAM_PAD == STREAM_PAD_RECS, "stream pad");

namespace hhv {
namespace api {
namespace {
thread_local std::string g_err;
}
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
const char* last_error() { return g_err.c_str(); }

static int error_word_check(hhv_ctx* c, const char* who);
int sync_check(hhv_ctx* c, const char* who) {
  const hipError_t e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(HHV_E_DEVICE, "%s: %s", who, hipGetErrorString(e));
  return error_word_check(c, who);
}
static int error_word_check(hhv_ctx* c, const char* who) {
  if (c->h_err) {
    const uint32_t w = *(volatile uint32_t*)c->h_err;
    if (w) {
      *(volatile uint32_t*)c->h_err = 0;
      return fail(HHV_E_DEVICE, "%s: device-side failure 0x%x:%s%s%s - the results of the launches since the last check are invalid", who, w,
                  (w & DEV_ERR_PAIR_TIMEOUT) ? " a wave of a two-wave workgroup waited in vain for its partner (pair kernel flow control)" : "",
                  (w & DEV_ERR_TRACE_STATE) ? " illegal state in the backtrace walk (src/hhviterbi.cpp:139-144)" : "",
                  (w & DEV_ERR_MAC_TIMEOUT) ? " a wavefront of a MAC forward / backward workgroup waited in vain for another one's progress (dataflow kernels)" : "");
    }
  }
  return HHV_OK;
}
int check_error_word(hhv_ctx* c, const char* who) { return error_word_check(c, who); }
}  // namespace api
}  // namespace hhv

namespace hhv {
size_t topk_temp_bytes(int n);
void results_to_hits(const DevResult* d_res, int n, DevHit* d_hits, hipStream_t stream);
int topk_device(const DevHit* d_hits, int n, int k, const int32_t* gids, DevHit* d_out, uint64_t* keys, uint64_t* sorted,
                void* temp, size_t temp_bytes, hipStream_t stream, std::string* err, const float* rank, const DevResult* d_results);
bool topk_small_enabled();  // HHV_TOPK_SMALL != 0
int topk_small_device(const DevHit* d_hits, const DevResult* d_results, int n, int k, const int32_t* gids, DevHit* d_out, hipStream_t stream,
                      std::string* err, const float* rank);
void topk_rank_pvalue(const DevHit* d_hits, int n, const int32_t* d_L, const float* d_neff, int Lq, float q_neff, int local, float* d_rank,
                      hipStream_t stream);
int merge_hits_device(const DevHit* d_in, int m, int k, DevHit* d_out, int* d_n, hipStream_t stream, std::string* err);
}

extern "C" {

int hhv_abi_version(void) { return HHV_ABI_VERSION; }
const char* hhv_last_error(void) { return hhv::api::last_error(); }

int32_t hhv_record_bytes(void) { return REC_DW * (int32_t)sizeof(float); }

int hhv_pack_profile(const float* p, const float* tr, int32_t L, int32_t index, float* out) {
  if (!p || !tr || !out || L < 1) return fail(HHV_E_ARG, "hhv_pack_profile: bad argument");
  const bool ok = index >= 0 ? pack_template(p, tr, L, index, out) : pack_columns(p, tr, L, out);
  if (!ok) return fail(HHV_E_ARG, "hhv_pack_profile: negative profile value (profile values are probabilities / odds)");
  return HHV_OK;
}

// src/util-inl.h:108-130: lg2[i] = log(float(1024+i))*1.442695041 - 10.0f, diff[i-1] = (lg2[i]-prev)*1.2352E-4.
// The reference initialises this table in whichever translation unit calls fast_log2 first; in a real run that
// is hhhmm.cpp (HMM::AddTransitionPseudocounts during PrepareQueryHMM), where log(float) binds to the DOUBLE
// log - not hhviterbi.cpp, where it would bind to logf.  The double flavour is therefore the one the Viterbi
// rescoring sees in hhsearch/hhblits, and the one built here (pinned through the oracle against oracle/_ref).
int hhv_fast_log2_tables(float* lg2, float* diff) {
  if (!lg2 || !diff) return fail(HHV_E_ARG, "hhv_fast_log2_tables: null");
  float prev = 0.0f;
  lg2[0] = 0.0f;
  diff[1024] = 0.0f;
  for (int i = 1; i <= 1024; ++i) {
    lg2[i] = (float)(log((double)(float)(1024 + i)) * 1.442695041 - (double)10.0f);
    diff[i - 1] = (float)((double)(lg2[i] - prev) * 1.2352E-4);
    prev = lg2[i];
  }
  return HHV_OK;
}

int hhv_device_count(int32_t* n) {
  if (!n) return fail(HHV_E_ARG, "hhv_device_count: null");
  int ndev = 0;
  const hipError_t e = hipGetDeviceCount(&ndev);
  *n = e == hipSuccess ? ndev : 0;
  if (e != hipSuccess || ndev <= 0)
    return fail(HHV_E_DEVICE, "no HIP device available (%s)", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  return HHV_OK;
}

// Segment table of the stream kernel's work queue (hhv_stream_kernel.h DQ): whole templates in stream order, a segment closed
// as soon as it holds >= HHV_SEGMENT_MIN_RECORDS records (a ring chunk of <= 32 records may hold one junction only, and a
// draw costs a round trip); a short remainder joins the segment in front of it.  Listed longest first (stable), so that what
// is left for the end of a launch are the short ones - a 1000-column template drawn last would add its whole length to the
// launch.  seg = (first record, end record) per segment in draw order + the terminal entry (total, total + 1).
static int plan_segments(const int64_t* rec_off, int n, std::vector<int64_t>& seg) {
  std::vector<int64_t> first;
  first.reserve((size_t)n + 1);
  int64_t start = 0;
  for (int k = 0; k < n; ++k) {
    if (rec_off[k + 1] - start >= HHV_SEGMENT_MIN_RECORDS) {
      first.push_back(start);
      start = rec_off[k + 1];
    }
  }
  const int64_t total = n > 0 ? rec_off[n] : 0;
  if (first.empty() && total > 0) first.push_back(0);  // (a remainder of < 128 records belongs to the last segment)
  const int n_seg = (int)first.size();
  first.push_back(total);
  std::vector<int32_t> order((size_t)n_seg);
  for (int k = 0; k < n_seg; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return first[x + 1] - first[x] > first[y + 1] - first[y]; });
  seg.assign((size_t)2 * (n_seg + 1), 0);
  for (int k = 0; k < n_seg; ++k) {
    seg[2 * (size_t)k] = first[order[k]];
    seg[2 * (size_t)k + 1] = first[order[k] + 1];
  }
  seg[2 * (size_t)n_seg] = total;  // the terminal header
  seg[2 * (size_t)n_seg + 1] = total + 1;
  return n_seg;
}

int hhv_segment_plan(int32_t n, const int32_t* L, int64_t* seg_out, int32_t* n_seg) {
  if (n < 0 || !n_seg || !seg_out || (n && !L)) return fail(HHV_E_ARG, "hhv_segment_plan: bad argument");
  std::vector<int64_t> rec_off((size_t)n + 1, 0);
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1) return fail(HHV_E_ARG, "hhv_segment_plan: template %d has %d columns", k, L[k]);
    rec_off[k + 1] = rec_off[k] + L[k] + 1;
  }
  std::vector<int64_t> seg;
  *n_seg = plan_segments(rec_off.data(), n, seg);
  std::copy(seg.begin(), seg.end(), seg_out);
  return HHV_OK;
}

int hhv_shard_plan(int32_t n, const int32_t* L, int32_t n_shards, int32_t* shard_of) {
  if (n < 0 || n_shards < 1 || (n && (!L || !shard_of))) return fail(HHV_E_ARG, "hhv_shard_plan: bad argument");
  if (n == 0) return HHV_OK;
  std::vector<int32_t> order((size_t)n);
  for (int k = 0; k < n; ++k) order[k] = k;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return L[a] > L[b]; });
  if (L[order[0]] == L[order[n - 1]]) {  // equal lengths: contiguous blocks (order == identity)
    for (int r = 0; r < n_shards; ++r)
      for (int64_t k = (int64_t)n * r / n_shards; k < (int64_t)n * (r + 1) / n_shards; ++k) shard_of[order[k]] = r;
    return HHV_OK;
  }
  const int bsz = std::max(1, std::min(64, n / (4 * n_shards)));  // 64-template bins, smaller for tiny databases
  std::vector<int64_t> load((size_t)n_shards, 0);
  for (int a = 0; a < n; a += bsz) {
    int r = 0;
    for (int t = 1; t < n_shards; ++t)
      if (load[t] < load[r]) r = t;
    for (int k = a; k < std::min(n, a + bsz); ++k) {
      shard_of[order[k]] = r;
      load[r] += (int64_t)L[order[k]] + 1;
    }
  }
  return HHV_OK;
}

int hhv_create(hhv_ctx** out, const hhv_params* par) {
  if (!out || !par) return fail(HHV_E_ARG, "hhv_create: null argument");
  *out = nullptr;
  // The MAC realignment runs its template-length classes side by side in MAC_CHAINS streams (hhv_mac.hip launch_mac).  The HIP
  // runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share one run one after the other:
  // 500 hits of mixed lengths 9.2 ms with eight queues, 12.7 ms with four (tools/SESSIONS.md round 6); with eight the context's own
  // stream, the null stream and the eight class streams still left two classes behind each other (4.9 -> 4.65 ms with sixteen).
  // The runtime reads the variable when it initialises, so this has an effect only in a process whose first HIP call is this one
  // (the drop-in applications); hosts that initialise HIP earlier export it themselves (pyhhv/capi.py does).  An explicit setting
  // is respected.
  (void)setenv("GPU_MAX_HW_QUEUES", "16", 0);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(HHV_E_DEVICE, "hhv_create: no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "0 devices" : hipGetErrorString(e));
  if (par->device < 0 || par->device >= ndev) return fail(HHV_E_ARG, "hhv_create: device %d of %d", par->device, ndev);
  HIP_TRY(hipSetDevice(par->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, par->device));
  hhv_ctx* c = new (std::nothrow) hhv_ctx();
  if (!c) return fail(HHV_E_MEMORY, "hhv_create: out of host memory");
  c->par = *par;
  c->num_cus = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_q, hipEventDisableTiming) != hipSuccess) {
    hhv_destroy(c);
    return fail(HHV_E_DEVICE, "hhv_create: stream/event creation failed");
  }
  if (hipHostMalloc((void**)&c->h_err, 2 * sizeof(uint32_t), hipHostMallocMapped) != hipSuccess ||  // [0] error word, [1] query upload done
      hipHostGetDevicePointer((void**)&c->d_err, c->h_err, 0) != hipSuccess) {
    hhv_destroy(c);
    return fail(HHV_E_DEVICE, "hhv_create: error word allocation failed");
  }
  c->h_err[0] = c->h_err[1] = 0;
  if (const char* e = getenv("HHV_QUERY_COPY")) c->q_by_copy = atoi(e) == 1;
  if (const char* e = getenv("HHV_EVENT_RECORDS")) c->ev_records = atoi(e) != 0;
  // launch-policy defaults from the environment, read once per context (hhv_set_launch_policy overrides them)
  if (const char* e = getenv("HHV_PAIR")) c->pair_mode = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("HHV_PAIR_SWAP")) c->pair_swap = std::max(-1, std::min(31, atoi(e)));
  if (const char* e = getenv("HHV_BLOCKS_PER_CU")) c->blocks_per_cu = std::max(0, atoi(e));
  if (const char* e = getenv("HHV_TRACE_WAVE")) c->trace_mode = atoi(e) != 0 ? 1 : 0;
  std::vector<float> lg2(1025), diff(1025);
  hhv_fast_log2_tables(lg2.data(), diff.data());
  if (hipMalloc(&c->d_lg2, 1025 * sizeof(float)) != hipSuccess ||
      hipMalloc(&c->d_diff, 1025 * sizeof(float)) != hipSuccess ||
      hipMemcpy(c->d_lg2, lg2.data(), 1025 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(c->d_diff, diff.data(), 1025 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
    hhv_destroy(c);
    return fail(HHV_E_DEVICE, "hhv_create: table upload failed");
  }
  {
    std::lock_guard<std::mutex> lock(g_live_m);
    g_live.push_back(c);
  }
  *out = c;
  return HHV_OK;
}

int hhv_set_fast_log2_tables(hhv_ctx* c, const float* lg2, const float* diff) {
  if (!c || !lg2 || !diff) return fail(HHV_E_ARG, "hhv_set_fast_log2_tables: null argument");
  HIP_TRY(hipSetDevice(c->par.device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpy(c->d_lg2, lg2, 1025 * sizeof(float), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(c->d_diff, diff, 1025 * sizeof(float), hipMemcpyHostToDevice));
  return HHV_OK;
}

int hhv_set_params(hhv_ctx* c, const hhv_params* par) {
  if (!c || !par) return fail(HHV_E_ARG, "hhv_set_params: null argument");
  if (par->device != c->par.device)
    return fail(HHV_E_ARG, "hhv_set_params: the context lives on device %d, not %d", c->par.device, par->device);
  c->par = *par;
  c->ss_dirty = true;  // the premultiplied table carries par.ssw
  return HHV_OK;
}

int hhv_set_launch_policy(hhv_ctx* c, int32_t pair_mode, int32_t pair_swap, int32_t blocks_per_cu, int32_t trace_mode) {
  if (!c) return fail(HHV_E_ARG, "hhv_set_launch_policy: null argument");
  if (pair_mode < -1 || pair_mode > 1) return fail(HHV_E_ARG, "hhv_set_launch_policy: pair_mode %d is not one of -1 (library's choice), 0 (one launch per strip), 1 (pairs wherever possible)", pair_mode);
  if (pair_swap < -1 || pair_swap > 31) return fail(HHV_E_ARG, "hhv_set_launch_policy: pair_swap %d outside [-1, 31]", pair_swap);
  if (blocks_per_cu < 0) return fail(HHV_E_ARG, "hhv_set_launch_policy: blocks_per_cu %d", blocks_per_cu);
  if (trace_mode < -1 || trace_mode > 1) return fail(HHV_E_ARG, "hhv_set_launch_policy: trace_mode %d is not one of -1, 0, 1", trace_mode);
  c->trace_mode = trace_mode;
  c->pair_mode = pair_mode;
  c->pair_swap = pair_swap;
  c->blocks_per_cu = blocks_per_cu;
  return HHV_OK;
}

void hhv_destroy(hhv_ctx* c) {
  if (!c) return;
  {
    std::lock_guard<std::mutex> lock(g_live_m);
    g_live.erase(std::remove(g_live.begin(), g_live.end(), c), g_live.end());
  }
  (void)hipSetDevice(c->par.device);
  dfree(c->d_qpack);  // (d_qp points into the same block)
  dfree(c->d_queue);
  dfree(c->d_lg2);
  dfree(c->d_diff);
  dfree(c->d_ss_table);
  dfree(c->d_ss_q_off);
  dfree(c->mac_cache);
  dfree(c->d_merge);
  for (int k = 0; k < MAC_CLASSES; ++k) {
    if (c->mac_side.s[k]) (void)hipStreamDestroy((hipStream_t)c->mac_side.s[k]);
    if (c->mac_side.join[k]) (void)hipEventDestroy((hipEvent_t)c->mac_side.join[k]);
  }
  if (c->mac_side.fork) (void)hipEventDestroy((hipEvent_t)c->mac_side.fork);
  hhv::api::pool_release(c);
  if (c->h_packed) (void)hipHostFree(c->h_packed);
  dfree(c->d_packed);
  if (c->mac_pinned) (void)hipHostFree(c->mac_pinned);
  if (c->mac_pinned_out) (void)hipHostFree(c->mac_pinned_out);
  if (c->q_stage) (void)hipHostFree(c->q_stage);
  if (c->ss_stage) (void)hipHostFree(c->ss_stage);
  if (c->co_stage) (void)hipHostFree(c->co_stage);
  dfree(c->d_co);
  if (c->ev_co) (void)hipEventDestroy(c->ev_co);
  if (c->ev_ss) (void)hipEventDestroy(c->ev_ss);
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->ev_q) (void)hipEventDestroy(c->ev_q);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // extern "C"
// The staging block of hhv_set_query (pinned host memory, device-visible) -> the context's device block, by ONE workgroup; the same
// launch sets the ticket counter of the stream kernel that follows (what a fill operation did) and, when everything has been read,
// tells the host through a host-mapped word that the staging block is free again (what an event did: on the stream of a search
// loop every event record and every copy / fill operation between two kernels costs 4 - 6 us, tools/trace10k.sh).
__global__ void __launch_bounds__(1024) query_upload_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int n4, uint32_t* __restrict__ queue,
                                                            uint32_t queue_value, uint32_t* __restrict__ done_flag, uint32_t seq) {
  // (the reads cross the bus: four per thread in flight - a query of 300 rows is 3.7 per thread)
  for (int base = 0; base < n4; base += 4096) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[min(base + u * 1024 + (int)threadIdx.x, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = base + u * 1024 + (int)threadIdx.x;
      if (i < n4) dst[i] = v[u];
    }
  }
  __syncthreads();  // (every thread's loads have returned: its stores depend on them)
  if (threadIdx.x == 0) {
    if (queue) *queue = queue_value;
    __hip_atomic_store(done_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
extern "C" {

int hhv_set_query(hhv_ctx* c, const float* p, const float* tr, int32_t Lq) {
  if (!c || !p || !tr) return fail(HHV_E_ARG, "hhv_set_query: null argument");
  if (Lq < 1) return fail(HHV_E_ARG, "hhv_set_query: Lq = %d", Lq);
  if (Lq > 0x7FFF) return fail(HHV_E_LIMIT, "hhv_set_query: Lq = %d exceeds 32767", Lq);
  // strips of 64*R rows, R <= 5 (keeps the kernel at 2 waves/SIMD): the fewest passes that cover Lq, rows spread evenly
  // (HHV_ARRAY_LANES = 64 / 32 keeps short queries on wider arrays: for measurements)
  const char* al = getenv("HHV_ARRAY_LANES");
  const StripPlan plan = StripPlan::make(Lq, al ? atoi(al) : 16);
  HIP_TRY(hipSetDevice(c->par.device));
  // The packed rows and the AoS profile go through ONE pinned staging block and two asynchronous copies on the context's
  // stream; device buffers and staging are kept between queries (grown when a longer query arrives).  A search loop that
  // sets a query per step therefore neither frees device memory (hipFree waits for the whole device) nor waits for the
  // stream: the only wait is for the PREVIOUS query's copies to have left the staging block (an event, long signalled).
  const size_t n_qpack = (size_t)plan.rows() * REC_DW, n_qp = (size_t)(Lq + 1) * 20;
  const size_t stage_bytes = (n_qpack + n_qp) * sizeof(float);
  c->Lq = 0;  // no query until the new one is in place (a failure below must not leave the old geometry with freed buffers)
  c->ss_dirty = true;
  if (c->q_stage_busy) {
    if (c->q_by_copy) {
      HIP_TRY(hipEventSynchronize(c->ev_q));
    } else {
      // the upload kernel of the previous query reports through h_err[1] (a stream that makes no progress for seconds: the runtime's wait)
      volatile uint32_t* flag = c->h_err + 1;
      for (long spin = 0; *flag != c->q_seq; ++spin) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
        if (spin > 100000000L) {
          HIP_TRY(hipStreamSynchronize(c->stream));
          break;
        }
      }
    }
    c->q_stage_busy = false;
  }
  if (c->q_stage_bytes < stage_bytes) {
    if (c->q_stage) (void)hipHostFree(c->q_stage);
    c->q_stage = nullptr;
    c->q_stage_bytes = 0;
    HIP_TRY(hipHostMalloc(&c->q_stage, stage_bytes, hipHostMallocDefault));
    c->q_stage_bytes = stage_bytes;
  }
  // ONE device block behind both arrays (the layout of the staging block), so that a query is one copy operation on the stream
  if (c->qpack_cap < n_qpack + n_qp) {
    HIP_TRY(hipStreamSynchronize(c->stream));  // (a launch that still reads the old rows)
    dfree(c->d_qpack);
    c->d_qp = nullptr;
    c->qpack_cap = 0;
    const size_t cap = n_qpack + n_qp + (n_qpack + n_qp) / 4;  // (a slightly longer query next time: no reallocation)
    HIP_TRY(hipMalloc(&c->d_qpack, cap * sizeof(float)));
    c->qpack_cap = cap;
  }
  c->d_qp = c->d_qpack + n_qpack;  // (n_qpack is a multiple of 64 * 28 floats: 16-byte aligned)
  float* const h_qpack = (float*)c->q_stage;
  float* const h_qp = h_qpack + n_qpack;
  memset(h_qpack, 0, n_qpack * sizeof(float));
  if (!pack_columns(p, tr, Lq, h_qpack))
    return fail(HHV_E_ARG, "hhv_set_query: negative profile value (profile values are probabilities)");
  memcpy(h_qp, p, n_qp * sizeof(float));
  // The block reaches the device through a KERNEL that reads the pinned staging block over the bus (query_upload_kernel), not
  // through a copy operation: on the stream of a search loop a copy sits between kernels, and the hand-over from the compute
  // queue to the copy engine and back costs more than moving 58 KB does (10 000-template step, tools/trace10k.sh: 26 us between
  // the merge kernel's end and the next launch with two hipMemcpyAsync).  The kernel is launched by the next hhv_align_async
  // (flush_query: the only readers of the block are that launch and the hhv_hits behind it), together with the ticket counter.
  // HHV_QUERY_COPY=1 (read when the context is created): the copy operation, at once.
  c->q_n4 = (n_qpack + n_qp) / 4;  // (both counts are multiples of four floats)
  if (c->q_by_copy) {
    HIP_TRY(hipMemcpyAsync(c->d_qpack, h_qpack, (n_qpack + n_qp) * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(c->ev_q, c->stream));
    c->q_stage_busy = true;
  } else {
    c->q_dirty = true;
  }
  c->Lq = Lq;
  c->plan = plan;
  c->q_pred.clear();
  c->q_conf.clear();
  c->q_dssp.clear();
  c->ss_dirty = true;
  return HHV_OK;
}

int hhv_set_ss_tables(hhv_ctx* c, const float* S73, const float* S33, const float* S37) {
  if (!c || !S73 || !S33 || !S37) return fail(HHV_E_ARG, "hhv_set_ss_tables: null argument");
  c->S73.assign(S73, S73 + 8 * 4 * 11);
  c->S33.assign(S33, S33 + 4 * 11 * 4 * 11);
  c->S37.assign(S37, S37 + 4 * 11 * 8);
  c->ss_dirty = true;
  return HHV_OK;
}

int hhv_set_query_ss(hhv_ctx* c, const int8_t* ss_pred, const int8_t* ss_conf, const int8_t* ss_dssp) {
  if (!c) return fail(HHV_E_ARG, "hhv_set_query_ss: null argument");
  if (c->Lq < 1) return fail(HHV_E_STATE, "hhv_set_query_ss: call hhv_set_query first");
  const size_t m = (size_t)c->Lq + 1;
  c->q_pred.clear();
  c->q_conf.clear();
  c->q_dssp.clear();
  if (ss_pred) c->q_pred.assign(ss_pred, ss_pred + m);
  if (ss_conf) c->q_conf.assign(ss_conf, ss_conf + m);
  if (ss_dssp) c->q_dssp.assign(ss_dssp, ss_dssp + m);
  c->ss_dirty = true;
  return HHV_OK;
}

int hhv_set_ss_mode(hhv_ctx* c, int32_t mode) {
  if (!c) return fail(HHV_E_ARG, "hhv_set_ss_mode: null argument");
  if (mode != 0 && mode != 1 && mode != 2 && mode != 4)
    return fail(HHV_E_ARG, "hhv_set_ss_mode: %d is not one of 0 (none), 1 (PRED_DSSP), 2 (DSSP_PRED), 4 (PRED_PRED)", mode);
  if (mode != 0 && c->S33.empty()) return fail(HHV_E_STATE, "hhv_set_ss_mode: call hhv_set_ss_tables first");
  c->ss_hmm_mode = mode;
  c->ss_dirty = true;
  return HHV_OK;
}

// (Re)build the device-side SS operands of the current mode: the premultiplied table ssw*S (the reference
// multiplies per cell, src/hhviterbialgorithm.cpp:209 - the same single fp32 product) and the per-row offsets
//   PRED_PRED: S33[q_pred][q_conf][t_pred][t_conf]  row (q_pred*11+q_conf)*44, column pred_index   (:199-200)
//   DSSP_PRED: S73[q_dssp][t_pred][t_conf]          row q_dssp*44,              column pred_index   (:201-202)
//   PRED_DSSP: S37[q_pred][q_conf][t_dssp]          row (q_pred*11+q_conf)*8,   column dssp_index   (:203-204)
static int ensure_ss(hhv_ctx* c) {
  if (!c->ss_dirty) return HHV_OK;
  // ss_dirty is cleared only once the operands are on their way to the device: after a failure the next call tries again
  // instead of launching the SS kernels with stale tables
  if (c->ss_hmm_mode == 0 || c->Lq < 1) {
    c->ss_dirty = false;
    return HHV_OK;
  }
  // A search loop sets a query - and with it the query's secondary structure - per search: like hhv_set_query this neither
  // frees device memory nor waits for the stream.  Table and offsets go through one pinned staging block with asynchronous
  // copies on the context's stream; the device buffers are kept (the table has a fixed capacity, the offsets grow with Lq).
  const std::vector<float>& T = c->ss_hmm_mode == 4 ? c->S33 : (c->ss_hmm_mode == 2 ? c->S73 : c->S37);
  const size_t rows = (size_t)c->plan.rows();
  constexpr size_t TAB_MAX = 4 * 11 * 4 * 11;
  if (T.size() > TAB_MAX) return fail(HHV_E_ARG, "secondary-structure table of %zu entries", T.size());
  const size_t stage_bytes = TAB_MAX * sizeof(float) + rows * sizeof(int32_t);
  if (c->ss_stage_busy) {
    HIP_TRY(hipEventSynchronize(c->ev_ss));
    c->ss_stage_busy = false;
  }
  if (c->ss_stage_bytes < stage_bytes) {
    if (c->ss_stage) (void)hipHostFree(c->ss_stage);
    c->ss_stage = nullptr;
    c->ss_stage_bytes = 0;
    HIP_TRY(hipHostMalloc(&c->ss_stage, stage_bytes, hipHostMallocDefault));
    c->ss_stage_bytes = stage_bytes;
  }
  if (!c->ev_ss) HIP_TRY(hipEventCreateWithFlags(&c->ev_ss, hipEventDisableTiming));
  if (!c->d_ss_table) HIP_TRY(hipMalloc(&c->d_ss_table, TAB_MAX * sizeof(float)));
  if (c->ss_q_cap < rows) {
    HIP_TRY(hipStreamSynchronize(c->stream));  // (a launch that still reads the old offsets)
    dfree(c->d_ss_q_off);
    c->ss_q_cap = 0;
    HIP_TRY(hipMalloc(&c->d_ss_q_off, rows * sizeof(int32_t)));
    c->ss_q_cap = rows;
  }
  float* const tab = (float*)c->ss_stage;
  int32_t* const off = (int32_t*)(tab + TAB_MAX);
  for (size_t k = 0; k < T.size(); ++k) tab[k] = c->par.ssw * T[k];
  memset(off, 0, rows * sizeof(int32_t));
  for (int i = 1; i <= c->Lq; ++i) {
    const int pred = c->q_pred.empty() ? 0 : (unsigned char)c->q_pred[i], conf = c->q_conf.empty() ? 0 : c->q_conf[i];
    const int dssp = c->q_dssp.empty() ? 0 : (unsigned char)c->q_dssp[i];
    int o;
    if (c->ss_hmm_mode == 4) o = (pred * 11 + conf) * 44;
    else if (c->ss_hmm_mode == 2) o = dssp * 44;
    else o = (pred * 11 + conf) * 8;
    if (o < 0 || (size_t)o + (c->ss_hmm_mode == 1 ? 8 : 44) > T.size())
      return fail(HHV_E_ARG, "query secondary-structure code out of range at row %d", i);
    off[i - 1] = o;
  }
  c->ss_t_shift = c->ss_hmm_mode == 1 ? META_DSSP_SHIFT : META_PRED_SHIFT;
  c->ss_t_mask = c->ss_hmm_mode == 1 ? META_DSSP_MASK : META_PRED_MASK;
  c->ss_tab_n = (int)T.size();
  HIP_TRY(hipMemcpyAsync(c->d_ss_table, tab, T.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->d_ss_q_off, off, rows * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->ev_ss, c->stream));
  c->ss_stage_busy = true;
  c->ss_dirty = false;
  return HHV_OK;
}

}  // extern "C"

hipError_t hhv::api::pool_malloc(hhv_ctx* c, void** p, size_t bytes) {
  *p = nullptr;
  const size_t want = (std::max<size_t>(bytes, 1) + 511) & ~(size_t)511;
  if (!c || !ctx_alive(c)) return hipMalloc(p, want);
  DevPool& pool = c->pool;
  {
    std::unique_lock<std::mutex> lock(pool.m);
    std::multimap<size_t, DevPool::Block>::iterator it = pool.free_blocks.lower_bound(want);
    if (it != pool.free_blocks.end() && it->first <= std::max(2 * want, want + ((size_t)64 << 10))) {
      const size_t have = it->first;
      const DevPool::Block b = it->second;
      pool.free_blocks.erase(it);
      pool.cached -= have;
      pool.live[b.p] = have;
      lock.unlock();
      if (b.ev) {
        (void)hipEventSynchronize(b.ev);  // whatever the previous owner had queued on the context's stream
        std::lock_guard<std::mutex> again(pool.m);
        pool.spare.push_back(b.ev);
      }
      *p = b.p;
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(p, want);
  if (e != hipSuccess) {  // give the cached blocks back and try once more
    (void)hipGetLastError();
    pool_release(c);
    e = hipMalloc(p, want);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(pool.m);
    pool.live[*p] = want;
  }
  return e;
}

void hhv::api::pool_free(hhv_ctx* c, void* p) {
  if (!p) return;
  if (!c || !ctx_alive(c)) {
    (void)hipFree(p);
    return;
  }
  DevPool& pool = c->pool;
  size_t bytes = 0;
  hipEvent_t ev = nullptr;
  {
    std::lock_guard<std::mutex> lock(pool.m);
    std::unordered_map<void*, size_t>::iterator it = pool.live.find(p);
    if (it != pool.live.end()) {
      bytes = it->second;
      pool.live.erase(it);
    }
    if (bytes != 0 && bytes <= POOL_BLOCK_MAX && pool.cached + bytes <= POOL_TOTAL_MAX) {
      if (!pool.spare.empty()) {
        ev = pool.spare.back();
        pool.spare.pop_back();
      }
    } else {
      bytes = 0;
    }
  }
  if (bytes == 0) {
    (void)hipFree(p);
    return;
  }
  if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
  if (!ev || hipEventRecord(ev, c->stream) != hipSuccess) {  // no event to order the next owner behind: not cached
    if (ev) (void)hipEventDestroy(ev);
    (void)hipFree(p);
    return;
  }
  std::lock_guard<std::mutex> lock(pool.m);
  pool.free_blocks.insert(std::make_pair(bytes, DevPool::Block{p, ev}));
  pool.cached += bytes;
}

void hhv::api::pool_release(hhv_ctx* c) {
  if (!c) return;
  DevPool& pool = c->pool;
  std::multimap<size_t, DevPool::Block> blocks;
  std::vector<hipEvent_t> spare;
  {
    std::lock_guard<std::mutex> lock(pool.m);
    blocks.swap(pool.free_blocks);
    spare.swap(pool.spare);
    pool.cached = 0;
  }
  for (std::multimap<size_t, DevPool::Block>::iterator it = blocks.begin(); it != blocks.end(); ++it) {
    if (it->second.ev) (void)hipEventDestroy(it->second.ev);
    (void)hipFree(it->second.p);
  }
  for (size_t k = 0; k < spare.size(); ++k) (void)hipEventDestroy(spare[k]);
}

int hhv::api::tset_init_common(hhv_ctx* c, hhv_tset* ts, int32_t n, const int32_t* L) {
  ts->ctx = c;
  ts->n = n;
  ts->L.assign(L, L + n);
  ts->rec_off.resize((size_t)n + 1);
  int64_t off = 0;
  for (int k = 0; k < n; ++k) {
    if (L[k] < 1) return fail(HHV_E_ARG, "template %d: L = %d out of range [1, 65535]", k, L[k]);
    if (L[k] > 0xFFFF) return fail(HHV_E_LIMIT, "template %d: L = %d exceeds 65535", k, L[k]);
    ts->rec_off[k] = off;
    off += (int64_t)L[k] + 1;
  }
  ts->rec_off[n] = off;
  ts->n_records = off + 1;
  HIP_TRY(tmalloc(ts->ctx, &ts->d_rec_off, (size_t)(n + 1) * sizeof(int64_t)));
  HIP_TRY(tmalloc(ts->ctx, &ts->d_L, (size_t)n * sizeof(int32_t)));
  HIP_TRY(tmalloc(ts->ctx, &ts->d_results, (size_t)n * sizeof(DevResult)));
  HIP_TRY(hipMemcpy(ts->d_rec_off, ts->rec_off.data(), (size_t)(n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(ts->d_L, ts->L.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  return HHV_OK;
}

extern "C" {

int hhv_upload_templates(hhv_ctx* c, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                         hhv_tset** out) {
  return hhv_upload_templates_ss(c, n, L, p, tr, nullptr, nullptr, nullptr, out);
}

int hhv_upload_templates_ss(hhv_ctx* c, int32_t n, const int32_t* L, const float* const* p, const float* const* tr,
                            const int8_t* const* ss_pred, const int8_t* const* ss_conf, const int8_t* const* ss_dssp,
                            hhv_tset** out) {
  if (!c || !L || !p || !tr || !out) return fail(HHV_E_ARG, "hhv_upload_templates: null argument");
  if (n < 1) return fail(HHV_E_ARG, "hhv_upload_templates: n = %d", n);
  *out = nullptr;
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_tset* ts = new (std::nothrow) hhv_tset();
  if (!ts) return fail(HHV_E_MEMORY, "out of host memory");
  int rc = tset_init_common(c, ts, n, L);
  if (rc != HHV_OK) {
    hhv_tset_free(ts);
    return rc;
  }
  const size_t total = (size_t)(ts->n_records + STREAM_PAD_RECS) * REC_DW;
  if (tmalloc(ts->ctx, &ts->d_records, total * sizeof(float)) != hipSuccess) {
    hhv_tset_free(ts);
    return fail(HHV_E_MEMORY, "hhv_upload_templates: device allocation of %zu bytes failed", total * sizeof(float));
  }
  ts->owns_records = true;
  // Pack and upload in slabs of <= 64 MiB: two pinned staging buffers used in turn - a slab is packed by a few host threads
  // (templates are independent) while the copy of the one before it is in flight.  (One thread into pageable memory + a
  // blocking copy: 12 GB/s, a quarter of what the link carries.)
  for (int t = 0; t < n; ++t) {
    if (!p[t] || !tr[t]) {
      hhv_tset_free(ts);
      return fail(HHV_E_ARG, "hhv_upload_templates: template %d has a null profile", t);
    }
  }
  const size_t slab_recs = (64u << 20) / (REC_DW * sizeof(float));
  size_t max_template = 0;
  for (int t = 0; t < n; ++t) max_template = std::max(max_template, (size_t)L[t] + 1);
  // (the staging follows the upload - a handful of templates does not pin 2 x 64 MiB; an upload that fits one slab
  // gets one buffer)
  const size_t upload_recs = (size_t)ts->n_records;
  const size_t stage_floats = std::max(std::min(slab_recs, upload_recs), max_template) * REC_DW;
  const int n_stage = upload_recs > slab_recs ? 2 : 1;
  float* stage[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  auto release = [&]() {
    for (int b = 0; b < 2; ++b) {
      if (done[b]) (void)hipEventDestroy(done[b]);
      if (stage[b]) (void)hipHostFree(stage[b]);
    }
  };
  for (int b = 0; b < n_stage; ++b) {
    if (hipHostMalloc(&stage[b], stage_floats * sizeof(float), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) {
      release();
      hhv_tset_free(ts);
      return fail(HHV_E_MEMORY, "hhv_upload_templates: pinned staging of %zu bytes failed", stage_floats * sizeof(float));
    }
  }
  int n_threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* e = getenv("HHV_PACK_THREADS")) n_threads = std::max(1, std::min(64, atoi(e)));
  int k = 0, slab = 0;
  std::atomic<int> bad_ss(-1), bad_neg(-1);
  while (k < n) {
    const int k0 = k;
    size_t recs = 0;
    while (k < n && (recs == 0 || recs + (size_t)L[k] + 1 <= slab_recs)) {
      recs += (size_t)L[k] + 1;
      ++k;
    }
    const int b = slab & 1;
    if (slab >= 2 && hipEventSynchronize(done[b]) != hipSuccess) {
      release();
      hhv_tset_free(ts);
      return fail(HHV_E_DEVICE, "hhv_upload_templates: H2D copy failed");
    }
    float* const buf = stage[b];
    const int64_t base = ts->rec_off[k0];
    auto pack_range = [&](int t0, int t1) {
      for (int t = t0; t < t1; ++t) {
        for (int j = 1; j <= L[t]; ++j) {
          const int pr = ss_pred && ss_pred[t] ? ss_pred[t][j] : 0, cf = ss_conf && ss_conf[t] ? ss_conf[t][j] : 0;
          const int ds = ss_dssp && ss_dssp[t] ? ss_dssp[t][j] : 0;
          if (pr < 0 || pr > 3 || cf < 0 || cf > 10 || ds < 0 || ds > 7) {
            int none = -1;
            bad_ss.compare_exchange_strong(none, t);
            return;
          }
        }
        if (!pack_template(p[t], tr[t], L[t], t, buf + (size_t)(ts->rec_off[t] - base) * REC_DW, ss_pred ? ss_pred[t] : nullptr,
                           ss_conf ? ss_conf[t] : nullptr, ss_dssp ? ss_dssp[t] : nullptr)) {
          int none = -1;
          bad_neg.compare_exchange_strong(none, t);
          return;
        }
      }
    };
    const int nt = std::max(1, std::min(n_threads, (k - k0) / 64));
    if (nt == 1) {
      pack_range(k0, k);
    } else {
      std::vector<std::thread> pool;
      int started = 0;
      try {
        for (; started < nt; ++started)
          pool.emplace_back(pack_range, k0 + (int)((int64_t)(k - k0) * started / nt), k0 + (int)((int64_t)(k - k0) * (started + 1) / nt));
      } catch (...) {  // no more threads to be had: the calling thread packs the shares that found none
      }
      for (int w = started; w < nt; ++w) pack_range(k0 + (int)((int64_t)(k - k0) * w / nt), k0 + (int)((int64_t)(k - k0) * (w + 1) / nt));
      for (auto& th : pool) th.join();
    }
    if (bad_ss.load() >= 0 || bad_neg.load() >= 0) {
      (void)hipStreamSynchronize(c->stream);
      release();
      hhv_tset_free(ts);
      if (bad_ss.load() >= 0) return fail(HHV_E_ARG, "template %d: secondary-structure code out of range", bad_ss.load());
      return fail(HHV_E_ARG, "hhv_upload_templates: template %d has a negative profile value (p = f / null model >= 0)", bad_neg.load());
    }
    if (hipMemcpyAsync(ts->d_records + (size_t)base * REC_DW, buf, recs * REC_DW * sizeof(float), hipMemcpyHostToDevice, c->stream) != hipSuccess ||
        hipEventRecord(done[b], c->stream) != hipSuccess) {
      (void)hipStreamSynchronize(c->stream);
      release();
      hhv_tset_free(ts);
      return fail(HHV_E_DEVICE, "hhv_upload_templates: H2D copy failed");
    }
    ++slab;
  }
  {
    const hipError_t e = hipStreamSynchronize(c->stream);
    release();
    if (e != hipSuccess) {
      hhv_tset_free(ts);
      return fail(HHV_E_DEVICE, "hhv_upload_templates: H2D copy failed");
    }
  }
  // terminal header + zeroed slack
  std::vector<float> tail((size_t)(1 + STREAM_PAD_RECS) * REC_DW, 0.0f);
  write_header(tail.data(), -1, 0);
  if (hipMemcpy(ts->d_records + (size_t)ts->rec_off[n] * REC_DW, tail.data(), tail.size() * sizeof(float),
                hipMemcpyHostToDevice) != hipSuccess) {
    hhv_tset_free(ts);
    return fail(HHV_E_DEVICE, "hhv_upload_templates: H2D copy failed");
  }
  *out = ts;
  return HHV_OK;
}

int hhv_adopt_device_stream(hhv_ctx* c, int32_t n, const int32_t* L, const void* d_records, hhv_tset** out) {
  if (!c || !L || !d_records || !out) return fail(HHV_E_ARG, "hhv_adopt_device_stream: null argument");
  if (n < 1) return fail(HHV_E_ARG, "hhv_adopt_device_stream: n = %d", n);
  *out = nullptr;
  HIP_TRY(hipSetDevice(c->par.device));
  hhv_tset* ts = new (std::nothrow) hhv_tset();
  if (!ts) return fail(HHV_E_MEMORY, "out of host memory");
  int rc = tset_init_common(c, ts, n, L);
  if (rc != HHV_OK) {
    hhv_tset_free(ts);
    return rc;
  }
  ts->d_records = (float*)d_records;
  ts->owns_records = false;
  *out = ts;
  return HHV_OK;
}


void hhv_tset_free(hhv_tset* ts) {
  if (!ts) return;
  if (hhv::api::context_alive(ts->ctx)) (void)hipSetDevice(ts->ctx->par.device);
  if (ts->owns_records) tfree(ts->ctx, ts->d_records);
  tfree(ts->ctx, ts->d_rec_off);
  tfree(ts->ctx, ts->d_L);
  tfree(ts->ctx, ts->d_results);
  tfree(ts->ctx, ts->d_wave_rec);
  tfree(ts->ctx, ts->d_seg);
  tfree(ts->ctx, ts->d_bt);
  tfree(ts->ctx, ts->d_carry);
  tfree(ts->ctx, ts->d_carry_mi);
  tfree(ts->ctx, ts->d_path_off);
  tfree(ts->ctx, ts->d_i_steps);
  tfree(ts->ctx, ts->d_j_steps);
  tfree(ts->ctx, ts->d_states);
  tfree(ts->ctx, ts->d_S);
  tfree(ts->ctx, ts->d_Sss);
  tfree(ts->ctx, ts->d_hits);
  tfree(ts->ctx, ts->d_topk);
  tfree(ts->ctx, ts->d_keys);
  tfree(ts->ctx, ts->d_sorted);
  tfree(ts->ctx, ts->d_sort_temp);
  tfree(ts->ctx, ts->d_raw_hits);
  tfree(ts->ctx, ts->d_gids);
  tfree(ts->ctx, ts->d_neff);
  tfree(ts->ctx, ts->d_rank);
  delete ts;
}

int32_t hhv_tset_size(const hhv_tset* ts) { return ts ? ts->n : 0; }
int64_t hhv_tset_records(const hhv_tset* ts) { return ts ? ts->n_records : 0; }
int64_t hhv_tset_cells(const hhv_tset* ts, int32_t Lq) {
  if (!ts) return 0;
  int64_t s = 0;
  for (int32_t l : ts->L) s += (int64_t)Lq * l;
  return s;
}

// Fixed ranges (the -DHHV_NO_QUEUE measurement build only; the product's arrays draw segments from a queue instead,
// ensure_segments): contiguous template ranges with ~equal record counts, cut at the first template boundary behind
// w x total / n.  All waves are resident at once (n_waves = CUs x blocks/CU the variant's VGPR/LDS budget admits).
// (The contiguous partition with the smallest maximum - bisection + greedy fill, largest range 1.4 % instead of 7.6 % over the
// mean on a 50..1000-column database - measured 1.3 % SLOWER: profiles/r3_ab.txt ab-r3-10.  The waves do not run at one speed,
// so the largest range is not what ends the launch; what fixes that is the queue.)
static int ensure_partition(hhv_ctx* c, hhv_tset* ts, int n_ranges, int n_slots) {
  // n_ranges ranges, padded with empty ones to n_slots (a wave of a short-query launch takes 64 / W ranges)
  if (ts->n_waves == n_ranges && ts->n_range_slots == n_slots && ts->d_wave_rec) return HHV_OK;
  tfree(ts->ctx, ts->d_wave_rec);
  std::vector<int64_t> wr((size_t)n_slots + 1);
  const int64_t total = ts->rec_off[ts->n];
  int k = 0;
  for (int w = 0; w < n_ranges; ++w) {
    const int64_t target = (int64_t)(((__int128)total * w) / n_ranges);
    while (k < ts->n && ts->rec_off[k] < target) ++k;
    wr[w] = ts->rec_off[k];
  }
  for (int w = n_ranges; w <= n_slots; ++w) wr[w] = total;
  HIP_TRY(tmalloc(ts->ctx, &ts->d_wave_rec, wr.size() * sizeof(int64_t)));
  HIP_TRY(hipMemcpyAsync(ts->d_wave_rec, wr.data(), wr.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  ts->n_waves = n_ranges;
  ts->n_range_slots = n_slots;
  return HHV_OK;
}

// the segment table on the device (plan_segments above; depends on the stream only) and the context's ticket counter
static int ensure_segments(hhv_ctx* c, hhv_tset* ts) {
  if (!c->d_queue) HIP_TRY(hipMalloc(&c->d_queue, sizeof(uint32_t)));
  if (ts->d_seg) return HHV_OK;
  std::vector<int64_t> seg;
  ts->n_seg = plan_segments(ts->rec_off.data(), ts->n, seg);
  HIP_TRY(tmalloc(ts->ctx, &ts->d_seg, seg.size() * sizeof(int64_t)));
  HIP_TRY(hipMemcpyAsync(ts->d_seg, seg.data(), seg.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HHV_OK;
}

static int ensure_bt(hhv_ctx* c, hhv_tset* ts) {
  if (ts->d_bt && ts->bt_plan == c->plan) return HHV_OK;
  tfree(ts->ctx, ts->d_bt);
  const size_t bytes = (size_t)c->plan.P * bt_plane_entries(ts->n_records, c->plan.W) * sizeof(uint64_t);
  if (tmalloc(ts->ctx, &ts->d_bt, bytes) != hipSuccess)
    return fail(HHV_E_MEMORY, "backtrace buffer of %zu bytes does not fit on the device", bytes);
  HIP_TRY(hipMemsetAsync(ts->d_bt, 0, bytes, c->stream));
  ts->bt_plan = c->plan;
  ts->bt_valid = false;
  ts->bt_dirty = false;
  return HHV_OK;
}

// hhv_set_query left the packed query in the pinned staging block: one launch moves it to the device (see there)
static int flush_query(hhv_ctx* c, uint32_t* d_queue, uint32_t queue_value) {
  c->q_seq += 1;
  hipLaunchKernelGGL(query_upload_kernel, dim3(1), dim3(1024), 0, c->stream, (const float4*)c->q_stage, (float4*)c->d_qpack, (int)c->q_n4,
                     d_queue, queue_value, c->d_err + 1, c->q_seq);
  HIP_TRY(hipGetLastError());
  c->q_dirty = false;
  c->q_stage_busy = true;
  return HHV_OK;
}

int hhv_align_async(hhv_ctx* c, hhv_tset* ts, uint32_t flags, void* d_out) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_align: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_align: template set belongs to another context");
  if (c->Lq < 1) return fail(HHV_E_STATE, "hhv_align: no query set");
  HIP_TRY(hipSetDevice(c->par.device));
  const bool celloff = (flags & HHV_ALIGN_CELLOFF) != 0;
  const bool bt = celloff || (flags & HHV_ALIGN_BACKTRACE) != 0;
  const bool local = c->par.local != 0;
  int rc = ensure_ss(c);
  if (rc != HHV_OK) return rc;
  // src/hhviterbi.cpp:175: the ...AndSS kernels run only for ssm == SCORE_ALIGNMENT and a non-zero ss_hmm_mode
  const bool ss = c->par.ss_mode == 2 && c->ss_hmm_mode != 0;
  // one wave partition for all passes: the smallest residency among the kernels of the plan (R_hi and R_hi - 1)
  const StripPlan& plan = c->plan;
  const bool multi = plan.P > 1;
  const int arrays = LANES / plan.W;  // systolic arrays per wave (short queries: 2 or 4), one stream range each
  int blocks_per_cu = 0;
  for (int R = plan.R(plan.P - 1); R <= plan.R_hi; ++R) {
    int nb = 0, vgprs = 0;
    rc = stream_kernel_occupancy(plan.W, R, local, bt, celloff, multi, ss, &nb, &vgprs);
    if (rc != 0 || nb < 1) return fail(HHV_E_DEVICE, "occupancy query failed (%d)", rc);
    blocks_per_cu = blocks_per_cu ? std::min(blocks_per_cu, nb) : nb;
  }
  if (c->blocks_per_cu >= 1) blocks_per_cu = std::min(blocks_per_cu, c->blocks_per_cu);  // measurement aid (hhv_set_launch_policy)
  const int n_ranges = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)c->num_cus * blocks_per_cu * arrays, ts->n));
  const int wpw = stream_kernel_waves(plan.W, ss);  // whole workgroups: hhv_ss_kernel has eight wavefronts (an array beyond the last range / segment returns at once)
  int n_waves = ((n_ranges + arrays - 1) / arrays + wpw - 1) / wpw * wpw;
  rc = ensure_partition(c, ts, n_ranges, n_waves * arrays);
  if (rc != HHV_OK) return rc;
  // the systolic arrays draw stream segments from a queue (hhv_stream_kernel.h DQ); the fixed ranges above serve the -DHHV_NO_QUEUE build
#if defined(HHV_NO_QUEUE)  // measurement build (matches hhv_stream_kernel.h): a fixed range per wave in every variant
  const bool queue = false;
#else
  const bool queue = true;
#endif
  if (queue) {
    rc = ensure_segments(c, ts);
    if (rc != HHV_OK) return rc;
    n_waves = (std::max(1, std::min(c->num_cus * blocks_per_cu * arrays, ts->n_seg)) + arrays - 1) / arrays;
  }
  n_waves = (n_waves + wpw - 1) / wpw * wpw;
  if (bt) {
    rc = ensure_bt(c, ts);
    if (rc != HHV_OK) return rc;
    if (celloff && ts->bt_dirty) {
      // The buffer still holds the compare bits of an earlier backtrace / cell-off launch and no mask has been set since
      // (hhv_set_celloff / hhv_set_celloff_paths clear it): read as masks they would switch off arbitrary cells.  The
      // reference's matrix has no cell switched off in that situation (src/hhviterbi.cpp:188), so neither has this launch.
      HIP_TRY(hipMemsetAsync(ts->d_bt, 0, (size_t)ts->bt_plan.P * bt_plane_entries(ts->n_records, ts->bt_plan.W) * sizeof(uint64_t), c->stream));
      ts->bt_dirty = false;
    }
  }
  StreamArgs a;
  a.records = ts->d_records;
  a.wave_rec = ts->d_wave_rec;
  a.qpack = c->d_qpack;
  a.results = d_out ? (DevResult*)d_out : ts->d_results;
  a.bt = ts->d_bt;
  a.negq = 0.0f - c->par.egq;  // (+0 for a zero penalty: viterbi_lane.h Params)
  a.negt = 0.0f - c->par.egt;
  a.shift = c->par.shift;
  a.Lq = c->Lq;
  a.carry = nullptr;
  a.carry_mi = nullptr;
  a.pair_swap = 0;
  a.bt_pass_stride = (int64_t)bt_plane_entries(ts->n_records, plan.W);
  a.ss_table = ss ? c->d_ss_table : nullptr;
  a.ss_q_off = ss ? c->d_ss_q_off : nullptr;
  a.ss_t_shift = c->ss_t_shift;
  a.ss_t_mask = c->ss_t_mask;
  a.ss_tab_n = c->ss_tab_n;
  a.seg_first = queue ? ts->d_seg : nullptr;
  a.n_seg = queue ? ts->n_seg : 0;
  a.queue = c->d_queue;
  if (multi) {
    if (!ts->d_carry) {
      HIP_TRY(tmalloc(ts->ctx, &ts->d_carry, (size_t)ts->n_records * sizeof(float4)));
      HIP_TRY(tmalloc(ts->ctx, &ts->d_carry_mi, (size_t)ts->n_records * sizeof(float)));
    }
    a.carry = ts->d_carry;
    a.carry_mi = ts->d_carry_mi;
  }
  // Two strips (321 .. 640 query rows): ONE launch of two-wave workgroups, the first strip's bottom row handed to the second
  // through LDS (hhv_stream_kernel.h PairLds) instead of two launches with the row in HBM.  Not for masked rounds, five-row ...AndSS
  // builds and five-row backtrace strips (LDS-parked query rows).  Score-only strips of UNEQUAL height (4 + 3, 5 + 4 rows per
  // lane) stay two launches: the lock step of a heavy and a light wave costs the pair more than the HBM carry costs the
  // launches since the first strip has kernels of its own (Lq 431: 12.13 vs 11.77 ms, profiles/r4_ab.txt ab-r4-7); with
  // backtrace the pair is ahead either way (15.05 vs 15.73).
  // More than two strips: a CHAIN of launches - neighbouring strips two by two as pair launches (HBM carries only between the
  // links: half the launches, half the rows through HBM), a strip without a partner or without a pair kernel as a launch of its
  // own.  hhv_set_launch_policy pair_mode 0 / 1: never / whenever a pair kernel exists (tests, measurements).
  const bool pairs_possible = queue && plan.P >= 2 && plan.W == LANES && !celloff && c->pair_mode != 0;
  const bool pairs_forced = c->pair_mode == 1;
  a.pair_swap = c->pair_swap;
  a.err = c->d_err;
  // the launch of the pass (or of the pair of passes) that starts at `pass`: pair workgroups (0 = a launch of its own), chain bits
  auto geometry = [&](int pass, int* chain, int* n_wg) {
    int pair_wgs = 0;
    *chain = 0;
    *n_wg = 0;
    if (pairs_possible && pass + 1 < plan.P) {
      *chain = (pass > 0 ? 1 : 0) | (pass + 2 < plan.P ? 2 : 0);
      const bool wanted = pairs_forced || plan.P > 2 || bt || plan.R(0) == plan.R(1);
      if (wanted) pair_wgs = pair_kernel_occupancy(plan.R(pass), plan.R(pass + 1), local, bt, *chain, ss);
    }
    if (pair_wgs > 0) {
      // (pairs = two-wave arrays; the ...AndSS kernels have four of them per workgroup: whole workgroups, a pair beyond the last
      // segment returns at once)
      const int ppw = pair_kernel_pairs_per_workgroup(ss);
      *n_wg = (std::max(1, std::min(c->num_cus * pair_wgs, ts->n_seg)) + ppw - 1) / ppw * ppw;
    }
    return pair_wgs;
  };
  // a query that is still in the staging block goes up now, and the same launch sets the first pass's ticket counter
  bool queue_set = false;
  if (c->q_dirty) {
    int chain0 = 0, n_wg0 = 0;
    const int pw0 = geometry(0, &chain0, &n_wg0);
    const bool has_q = pw0 > 0 || queue;
    rc = flush_query(c, has_q ? c->d_queue : nullptr, pw0 > 0 ? (uint32_t)n_wg0 : (uint32_t)(n_waves * arrays));
    if (rc != HHV_OK) return rc;
    queue_set = has_q;
  }
  // the DP's time (hhv_last_kernel_ms): event records in front of and behind the launches.  HHV_EVENT_RECORDS=0 (read when the
  // context is created) attaches the two events to the first and the last launch's own dispatch instead (hipExtLaunchKernel) - measured
  // in one session (tools/evvar.sh): the same kernel times, the step of a 10 000-template search 1.4789 -> 1.4754 ms; a dispatch
  // that carries events pays for them much as the records do, so the long-tested form stays the default
  if (c->ev_records) HIP_TRY(hipEventRecord(c->ev0, c->stream));
  for (int pass = 0; pass < plan.P;) {
    a.row_base = plan.base(pass);
    a.bt_plane = pass;
    a.qpack = c->d_qpack + (size_t)a.row_base * REC_DW;
    a.pass_first = pass == 0;
    int chain = 0, n_wg = 0;
    const int pair_wgs = geometry(pass, &chain, &n_wg);
    const bool set_here = !(pass == 0 && queue_set);  // (pass 0: flush_query may have set the counter)
    if (pair_wgs > 0) {
      a.pass_last = 0;  // (the kernel gives its two waves their own)
      if (set_here) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->d_queue, n_wg, 1, c->stream));  // pair k starts with segment k
      rc = launch_pair(plan.R(pass), plan.R(pass + 1), local, bt, chain, ss, a, n_wg, c->stream, (!c->ev_records && pass == 0) ? c->ev0 : nullptr,
                       (!c->ev_records && pass + 2 >= plan.P) ? c->ev1 : nullptr);
      pass += 2;
    } else {
      a.pass_last = pass == plan.P - 1;
      if (queue && set_here) HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c->d_queue, n_waves * arrays, 1, c->stream));  // array k starts with segment k
      rc = launch_stream(plan.W, plan.R(pass), local, bt, celloff, multi, ss, a, n_waves, c->stream, (!c->ev_records && pass == 0) ? c->ev0 : nullptr,
                         (!c->ev_records && pass + 1 >= plan.P) ? c->ev1 : nullptr);
      pass += 1;
    }
    if (rc != 0) return fail(HHV_E_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc)));
  }
  if (c->ev_records) HIP_TRY(hipEventRecord(c->ev1, c->stream));
  c->ev_valid = true;
  if (d_out) {
    HIP_TRY(hipMemcpyAsync(ts->d_results, d_out, (size_t)ts->n * sizeof(DevResult), hipMemcpyDeviceToDevice, c->stream));
  }
  if (bt) {
    ts->bt_valid = true;
    ts->bt_dirty = true;
    ts->bt_Lq = c->Lq;
    ts->bt_mm = bt_mm_mode(plan.W, plan.R_hi, local, celloff, ss);  // (multi-pass plans are 64-lane plans: one encoding for all passes)
  }
  ts->hits_valid = false;
  return HHV_OK;
}

int hhv_sync(hhv_ctx* c) {
  if (!c) return fail(HHV_E_ARG, "hhv_sync: null");
  return sync_check(c, "hhv_sync");
}

int hhv_check_error(hhv_ctx* c) {
  if (!c) return fail(HHV_E_ARG, "hhv_check_error: null context");
  return hhv::api::check_error_word(c, "hhv_check_error");
}

void* hhv_stream(hhv_ctx* c) { return c ? (void*)c->stream : nullptr; }

int hhv_last_kernel_ms(hhv_ctx* c, float* ms) {
  if (!c || !ms) return fail(HHV_E_ARG, "hhv_last_kernel_ms: null");
  if (!c->ev_valid) return fail(HHV_E_STATE, "hhv_last_kernel_ms: no launch yet");
  HIP_TRY(hipEventSynchronize(c->ev1));
  HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return HHV_OK;
}

int hhv_align(hhv_ctx* c, hhv_tset* ts, uint32_t flags, hhv_result* out) {
  int rc = hhv_align_async(c, ts, flags, nullptr);
  if (rc != HHV_OK) return rc;
  if (out) {
    HIP_TRY(hipMemcpyAsync(out, ts->d_results, (size_t)ts->n * sizeof(DevResult), hipMemcpyDeviceToHost, c->stream));
  }
  return sync_check(c, "hhv_align");
}

int hhv_set_celloff(hhv_ctx* c, hhv_tset* ts, int32_t k, const uint8_t* mask) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_set_celloff: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_set_celloff: template set belongs to another context");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_set_celloff: template %d of %d", k, ts->n);
  if (c->Lq < 1) return fail(HHV_E_STATE, "hhv_set_celloff: no query set");
  HIP_TRY(hipSetDevice(c->par.device));
  int rc = ensure_bt(c, ts);
  if (rc != HHV_OK) return rc;
  if (ts->bt_dirty) {
    // an earlier backtrace / cell-off launch left compare bits in every entry; as masks they are garbage for the
    // templates that do not get a fresh mask now: start from "no cell excluded" for the whole set
    HIP_TRY(hipMemsetAsync(ts->d_bt, 0, (size_t)ts->bt_plan.P * bt_plane_entries(ts->n_records, ts->bt_plan.W) * sizeof(uint64_t), c->stream));
    ts->bt_dirty = false;
  }
  // the mask bytes go to the device once; a kernel turns them into the entries of template k (bit 7 of byte r of the
  // entry of (column j, lane): the only input bit of the kernel in this buffer)
  const int Lt = ts->L[k], Lq = c->Lq;
  unsigned char* d_mask = nullptr;
  if (mask) {
    const size_t bytes = (size_t)(Lq + 1) * (Lt + 1);
    HIP_TRY(tmalloc(c, &d_mask, bytes));
    if (hipMemcpyAsync(d_mask, mask, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      tfree(c, d_mask);
      return fail(HHV_E_DEVICE, "hhv_set_celloff: H2D copy failed");
    }
  }
  const int lr = celloff_from_mask(ts->d_bt, ts->d_rec_off, ts->d_L, (int64_t)bt_plane_entries(ts->n_records, c->plan.W), Lq,
                                   c->plan, k, d_mask, Lt, c->stream);
  const hipError_t e = hipStreamSynchronize(c->stream);
  tfree(c, d_mask);
  if (lr != 0 || e != hipSuccess) return fail(HHV_E_DEVICE, "hhv_set_celloff: device operation failed");
  ts->bt_valid = false;
  return HHV_OK;
}

int hhv_set_global_batch(hhv_ctx* c, hhv_tset* ts, const uint8_t* not_longest) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_set_global_batch: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_set_global_batch: template set belongs to another context");
  if (!ts->owns_records) return fail(HHV_E_STATE, "hhv_set_global_batch: the set uses an adopted stream buffer, which is not modified");
  HIP_TRY(hipSetDevice(c->par.device));
  unsigned char* d_flags = nullptr;
  if (not_longest) {
    HIP_TRY(tmalloc(c, &d_flags, (size_t)ts->n));
    if (hipMemcpyAsync(d_flags, not_longest, (size_t)ts->n, hipMemcpyHostToDevice, c->stream) != hipSuccess) {
      tfree(c, d_flags);
      return fail(HHV_E_DEVICE, "hhv_set_global_batch: H2D copy failed");
    }
  }
  const int lr = set_header_flags(ts->d_records, ts->d_rec_off, d_flags, ts->n, c->stream);
  const hipError_t e = hipStreamSynchronize(c->stream);
  tfree(c, d_flags);
  if (lr != 0 || e != hipSuccess) return fail(HHV_E_DEVICE, "hhv_set_global_batch: kernel failed");
  return HHV_OK;
}

int hhv_set_celloff_paths(hhv_ctx* c, hhv_tset* ts, int32_t n_paths, const int32_t* template_of, const int64_t* path_off,
                          const int32_t* i_steps, const int32_t* j_steps, int32_t n_qranges, const int32_t* qranges,
                          int32_t n_tranges, const int32_t* tranges) {
  if (!c || !ts || n_paths < 0 || n_qranges < 0 || n_tranges < 0 || (n_paths && (!template_of || !path_off || !i_steps || !j_steps)) ||
      (n_qranges && !qranges) || (n_tranges && !tranges))
    return fail(HHV_E_ARG, "hhv_set_celloff_paths: bad argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_set_celloff_paths: template set belongs to another context");
  if (c->Lq < 1) return fail(HHV_E_STATE, "hhv_set_celloff_paths: no query set");
  for (int r = 0; r < n_qranges + n_tranges; ++r) {
    const int32_t* rg = r < n_qranges ? qranges + 2 * r : tranges + 2 * (r - n_qranges);
    if (rg[0] < 1) return fail(HHV_E_ARG, "hhv_set_celloff_paths: range %d starts at %d (rows and columns are 1-based)", r, rg[0]);
  }
  for (int p = 0; p < n_paths; ++p)
    if (template_of[p] < 0 || template_of[p] >= ts->n || path_off[p + 1] < path_off[p])
      return fail(HHV_E_ARG, "hhv_set_celloff_paths: path %d invalid", p);
  HIP_TRY(hipSetDevice(c->par.device));
  int rc = ensure_bt(c, ts);
  if (rc != HHV_OK) return rc;
  const int64_t steps = n_paths ? path_off[n_paths] : 0;
  std::vector<int32_t> ranges;
  for (int r = 0; r < 2 * n_qranges; ++r) ranges.push_back(qranges[r]);
  for (int r = 0; r < 2 * n_tranges; ++r) ranges.push_back(tranges[r]);
  ranges.push_back(0);
  // One device block: template_of | path_off | i | j | ranges - and one pinned staging block of the same layout.  Both live in the
  // context and only ever grow (round 5): an alternative-alignment round used to pay a hipMalloc, four copies out of pageable
  // memory, a stream synchronisation and a hipFree (which waits for the whole device) per call - 10 ms for the 1.5 M path steps of
  // 10 000 templates, four times the masked DP that follows.  Now the caller's arrays are copied into the staging block (the
  // caller may reuse them when the call returns), the copies and the two kernels are enqueued on the context's stream and the
  // call returns; the staging block is guarded by an event like the query's.
  const size_t b_t = (size_t)n_paths * 4, b_o = (size_t)(n_paths + 1) * 8, b_s = (size_t)steps * 4, b_r = ranges.size() * 4;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t o_t = 0, o_o = o_t + al(b_t), o_i = o_o + al(b_o), o_j = o_i + al(b_s), o_r = o_j + al(b_s), total = o_r + al(b_r) + 256;
  if (c->co_busy) {
    HIP_TRY(hipEventSynchronize(c->ev_co));
    c->co_busy = false;
  }
  if (!c->ev_co) HIP_TRY(hipEventCreateWithFlags(&c->ev_co, hipEventDisableTiming));
  if (c->co_bytes < total) {
    HIP_TRY(hipStreamSynchronize(c->stream));  // (kernels that still read the old block)
    if (c->co_stage) (void)hipHostFree(c->co_stage);
    dfree(c->d_co);
    c->co_stage = nullptr;
    c->co_bytes = 0;
    const size_t cap = total + total / 4;
    HIP_TRY(hipHostMalloc(&c->co_stage, cap, hipHostMallocDefault));
    if (hipMalloc(&c->d_co, cap) != hipSuccess) {
      (void)hipHostFree(c->co_stage);
      c->co_stage = nullptr;
      return fail(HHV_E_MEMORY, "hhv_set_celloff_paths: %zu bytes of device scratch", cap);
    }
    c->co_bytes = cap;
  }
  char* const h = (char*)c->co_stage;
  char* const d = (char*)c->d_co;
  memcpy(h + o_r, ranges.data(), b_r);
  if (n_paths) {
    memcpy(h + o_t, template_of, b_t);
    memcpy(h + o_o, path_off, b_o);
    memcpy(h + o_i, i_steps, b_s);
    memcpy(h + o_j, j_steps, b_s);
  }
  HIP_TRY(hipMemcpyAsync(d, h, o_r + b_r, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->ev_co, c->stream));
  c->co_busy = true;
  const int lr = celloff_from_paths(ts->d_bt, ts->d_rec_off, ts->d_L, (int64_t)bt_plane_entries(ts->n_records, c->plan.W), c->Lq, c->plan, ts->n,
                                    n_paths, (const int32_t*)(d + o_t), (const int64_t*)(d + o_o), (const int32_t*)(d + o_i),
                                    (const int32_t*)(d + o_j), (const int32_t*)(d + o_r), n_qranges, n_tranges,
                                    ts->n ? *std::max_element(ts->L.begin(), ts->L.end()) : 0, c->stream);
  if (lr != 0) return fail(HHV_E_DEVICE, "hhv_set_celloff_paths: kernel launch failed");
  ts->bt_valid = false;
  ts->bt_dirty = false;  // the clear kernel rewrote every entry of every template
  return HHV_OK;
}

int hhv_backtrace_matrix(hhv_ctx* c, hhv_tset* ts, int32_t k, uint8_t* out) {
  if (!c || !ts || !out) return fail(HHV_E_ARG, "hhv_backtrace_matrix: null argument");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_backtrace_matrix: template %d of %d", k, ts->n);
  if (!ts->bt_valid || !ts->d_bt) return fail(HHV_E_STATE, "hhv_backtrace_matrix: no backtrace computed");
  HIP_TRY(hipSetDevice(c->par.device));
  const int Lt = ts->L[k], Lq = ts->bt_Lq;
  const size_t bytes = (size_t)(Lq + 1) * (Lt + 1);
  unsigned char* d_out = nullptr;
  HIP_TRY(tmalloc(c, &d_out, bytes));
  const int lr = bt_matrix(ts->d_bt, ts->d_rec_off, (int64_t)bt_plane_entries(ts->n_records, ts->bt_plan.W), Lq, ts->bt_plan, ts->bt_mm, k, Lt,
                           d_out, c->stream);
  hipError_t e = lr == 0 ? hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream) : hipErrorUnknown;
  const int sc = e == hipSuccess ? sync_check(c, "hhv_backtrace_matrix") : HHV_OK;
  tfree(c, d_out);
  if (e != hipSuccess) return fail(HHV_E_DEVICE, "hhv_backtrace_matrix: device operation failed");
  return sc;
}

static int ensure_paths(hhv_ctx* c, hhv_tset* ts) {
  if (ts->path_Lq == c->Lq && ts->d_hits) return HHV_OK;
  tfree(ts->ctx, ts->d_path_off);
  tfree(ts->ctx, ts->d_i_steps);
  tfree(ts->ctx, ts->d_j_steps);
  tfree(ts->ctx, ts->d_states);
  tfree(ts->ctx, ts->d_S);
  tfree(ts->ctx, ts->d_Sss);
  tfree(ts->ctx, ts->d_hits);
  ts->path_off.resize((size_t)ts->n + 1);
  int64_t off = 0;
  for (int k = 0; k < ts->n; ++k) {
    ts->path_off[k] = off;
    // BacktraceResult arrays: i2 + j2 + 2 entries (src/hhviterbi.cpp:90-93); pools start on multiples of four entries (the
    // trace kernel writes its state bytes in words of four)
    off += ((int64_t)c->Lq + ts->L[k] + 2 + 3) & ~(int64_t)3;
  }
  ts->path_off[ts->n] = off;
  HIP_TRY(tmalloc(ts->ctx, &ts->d_path_off, ts->path_off.size() * sizeof(int64_t)));
  HIP_TRY(tmalloc(ts->ctx, &ts->d_i_steps, (size_t)off * sizeof(int32_t)));
  HIP_TRY(tmalloc(ts->ctx, &ts->d_j_steps, (size_t)off * sizeof(int32_t)));
  HIP_TRY(tmalloc(ts->ctx, &ts->d_states, (size_t)off * sizeof(int8_t)));
  HIP_TRY(tmalloc(ts->ctx, &ts->d_S, ((size_t)off + 64) * sizeof(float)));  // (+ 64: hhv_scorr_kernel reads whole 64-step tiles)
  HIP_TRY(tmalloc(ts->ctx, &ts->d_hits, (size_t)ts->n * sizeof(DevHit)));
  HIP_TRY(hipMemcpy(ts->d_path_off, ts->path_off.data(), ts->path_off.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  ts->path_Lq = c->Lq;
  return HHV_OK;
}

static int run_trace(hhv_ctx* c, hhv_tset* ts) {
  if (!ts->bt_valid || !ts->d_bt || ts->bt_Lq != c->Lq)
    return fail(HHV_E_STATE, "hhv_hits: needs a preceding hhv_align with HHV_ALIGN_BACKTRACE for the current query");
  int rc = ensure_paths(c, ts);
  if (rc != HHV_OK) return rc;
  TraceArgs a;
  a.records = ts->d_records;
  a.rec_off = ts->d_rec_off;
  a.L = ts->d_L;
  a.qp = c->d_qp;
  a.results = ts->d_results;
  a.bt = ts->d_bt;
  a.lg2 = c->d_lg2;
  a.diff = c->d_diff;
  a.hits = ts->d_hits;
  a.i_steps = ts->d_i_steps;
  a.j_steps = ts->d_j_steps;
  a.states = ts->d_states;
  a.S = ts->d_S;
  a.path_off = ts->d_path_off;
  a.corr = c->par.corr;
  a.ss_mode = c->par.ss_mode;
  a.Lq = c->Lq;
  a.plan = ts->bt_plan;
  a.n = ts->n;
  a.bt_pass_stride = (int64_t)bt_plane_entries(ts->n_records, ts->bt_plan.W);
  a.bt_mm = ts->bt_mm;
  rc = ensure_ss(c);
  if (rc != HHV_OK) return rc;
  a.ss_table = c->ss_hmm_mode ? c->d_ss_table : nullptr;
  a.ss_q_off = c->ss_hmm_mode ? c->d_ss_q_off : nullptr;
  a.ss_t_shift = c->ss_t_shift;
  a.ss_t_mask = c->ss_t_mask;
  a.Sss = nullptr;
  if (c->ss_hmm_mode) {  // the per-step secondary-structure scores: a second pool like S, allocated with the first search that has them
    if (!ts->d_Sss) HIP_TRY(tmalloc(ts->ctx, &ts->d_Sss, ((size_t)ts->path_off[ts->n] + 64) * sizeof(float)));
    a.Sss = ts->d_Sss;
  }
  a.err = c->d_err;
  a.trace_mode = c->trace_mode;
  rc = launch_trace(a, c->stream);
  if (rc != 0) return fail(HHV_E_DEVICE, "trace kernel launch failed: %s", hipGetErrorString((hipError_t)(-rc)));
  ts->hits_valid = true;
  ts->host_paths_valid = false;
  ts->packed_valid = false;
  return HHV_OK;
}

int hhv_hits(hhv_ctx* c, hhv_tset* ts, hhv_hit* hits) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_hits: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_hits: template set belongs to another context");
  HIP_TRY(hipSetDevice(c->par.device));
  int rc = run_trace(c, ts);
  if (rc != HHV_OK) return rc;
  if (hits) {  // (hits == NULL: trace and rescoring are only enqueued on the context's stream, like hhv_align_async)
    HIP_TRY(hipMemcpyAsync(hits, ts->d_hits, (size_t)ts->n * sizeof(DevHit), hipMemcpyDeviceToHost, c->stream));
    return sync_check(c, "hhv_hits");
  }
  return HHV_OK;
}

int hhv_hit_path(hhv_ctx* c, hhv_tset* ts, int32_t k, int32_t cap, int32_t* i_steps, int32_t* j_steps, int8_t* states,
                 float* S, int32_t* nsteps) {
  if (!c || !ts || !nsteps) return fail(HHV_E_ARG, "hhv_hit_path: null argument");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_hit_path: template %d of %d", k, ts->n);
  if (!ts->hits_valid) return fail(HHV_E_STATE, "hhv_hit_path: call hhv_hits first");
  HIP_TRY(hipSetDevice(c->par.device));
  {
    const int src_ = sync_check(c, "hhv_hit_path");
    if (src_ != HHV_OK) return src_;
  }
  const int64_t pool = ts->path_off[ts->n];
  if (!ts->host_paths_valid && (size_t)pool * 13 <= ((size_t)768 << 20)) {
    ts->h_i_steps.resize((size_t)pool);
    ts->h_j_steps.resize((size_t)pool);
    ts->h_states.resize((size_t)pool);
    ts->h_S.resize((size_t)pool);
    ts->h_hits.resize((size_t)ts->n);
    HIP_TRY(hipMemcpy(ts->h_i_steps.data(), ts->d_i_steps, (size_t)pool * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ts->h_j_steps.data(), ts->d_j_steps, (size_t)pool * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ts->h_states.data(), ts->d_states, (size_t)pool * sizeof(int8_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ts->h_S.data(), ts->d_S, (size_t)pool * sizeof(float), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ts->h_hits.data(), ts->d_hits, (size_t)ts->n * sizeof(DevHit), hipMemcpyDeviceToHost));
    ts->host_paths_valid = true;
  }
  const int64_t po = ts->path_off[k];
  if (ts->host_paths_valid) {
    const int ns = ts->h_hits[k].nsteps;
    *nsteps = ns;
    const int m = std::min(cap, ns + 1);
    if (m <= 0) return HHV_OK;
    if (i_steps) memcpy(i_steps, ts->h_i_steps.data() + po, (size_t)m * sizeof(int32_t));
    if (j_steps) memcpy(j_steps, ts->h_j_steps.data() + po, (size_t)m * sizeof(int32_t));
    if (states) memcpy(states, ts->h_states.data() + po, (size_t)m * sizeof(int8_t));
    if (S) memcpy(S, ts->h_S.data() + po, (size_t)m * sizeof(float));
  } else {  // a path pool too large to mirror on the host: copy this hit only
    DevHit h;
    HIP_TRY(hipMemcpy(&h, ts->d_hits + k, sizeof(h), hipMemcpyDeviceToHost));
    *nsteps = h.nsteps;
    const int m = std::min(cap, h.nsteps + 1);
    if (m <= 0) return HHV_OK;
    if (i_steps) HIP_TRY(hipMemcpy(i_steps, ts->d_i_steps + po, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (j_steps) HIP_TRY(hipMemcpy(j_steps, ts->d_j_steps + po, (size_t)m * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (states) HIP_TRY(hipMemcpy(states, ts->d_states + po, (size_t)m * sizeof(int8_t), hipMemcpyDeviceToHost));
    if (S) HIP_TRY(hipMemcpy(S, ts->d_S + po, (size_t)m * sizeof(float), hipMemcpyDeviceToHost));
  }
  if (i_steps) i_steps[0] = 0;
  if (j_steps) j_steps[0] = 0;
  if (states) states[0] = 0;
  if (S) S[0] = 0.0f;
  return HHV_OK;
}

// SURVEY.md 8(b)'s hhv_backtrace: Viterbi::Backtrace for ONE template as the reference's caller sees it (BacktraceResult:
// i_steps / j_steps / states, count, matched_cols).  The walk itself runs for all templates of the set at once on the device
// (run_trace, started here if hhv_hits has not been called since the last backtrace launch); this entry hands out one result.
int hhv_backtrace(hhv_ctx* c, hhv_tset* ts, int32_t k, int32_t cap, int32_t* i_steps, int32_t* j_steps, int8_t* states,
                  int32_t* nsteps, int32_t* matched_cols) {
  if (!c || !ts || !nsteps) return fail(HHV_E_ARG, "hhv_backtrace: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_backtrace: template set belongs to another context");
  if (k < 0 || k >= ts->n) return fail(HHV_E_ARG, "hhv_backtrace: template %d of %d", k, ts->n);
  if (!ts->hits_valid) {
    const int rc = hhv_hits(c, ts, nullptr);
    if (rc != HHV_OK) return rc;
  }
  const int rc = hhv_hit_path(c, ts, k, cap, i_steps, j_steps, states, nullptr, nsteps);
  if (rc != HHV_OK) return rc;
  if (matched_cols) {
    if (ts->host_paths_valid) {
      *matched_cols = ts->h_hits[k].matched_cols;
    } else {
      DevHit h;
      HIP_TRY(hipMemcpy(&h, ts->d_hits + k, sizeof(h), hipMemcpyDeviceToHost));
      *matched_cols = h.matched_cols;
    }
  }
  return HHV_OK;
}

int hhv_hit_path_pool(hhv_ctx* c, hhv_tset* ts, const int64_t** path_off, const int32_t** i_steps, const int32_t** j_steps,
                      const int8_t** states, const float** S) {
  if (!c || !ts || !path_off || !i_steps || !j_steps || !states || !S) return fail(HHV_E_ARG, "hhv_hit_path_pool: null argument");
  if (!ts->hits_valid) return fail(HHV_E_STATE, "hhv_hit_path_pool: call hhv_hits first");
  int32_t ns = 0;
  const int rc = hhv_hit_path(c, ts, 0, 0, nullptr, nullptr, nullptr, nullptr, &ns);  // mirrors the pool on first use
  if (rc != HHV_OK) return rc;
  if (!ts->host_paths_valid) return fail(HHV_E_LIMIT, "hhv_hit_path_pool: the path pool is too large to mirror; use hhv_hit_path");
  *path_off = ts->path_off.data();
  *i_steps = ts->h_i_steps.data();
  *j_steps = ts->h_j_steps.data();
  *states = ts->h_states.data();
  *S = ts->h_S.data();
  return HHV_OK;
}

int hhv_hit_paths_packed(hhv_ctx* c, hhv_tset* ts, const hhv_hit* hits, const int64_t** off, const uint16_t** i_steps,
                         const uint16_t** j_steps, const int8_t** states, const float** S) {
  if (!c || !ts || !hits || !off || !i_steps || !j_steps || !states || !S) return fail(HHV_E_ARG, "hhv_hit_paths_packed: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_hit_paths_packed: template set belongs to another context");
  if (!ts->hits_valid) return fail(HHV_E_STATE, "hhv_hit_paths_packed: call hhv_hits first");
  if (c->Lq > 0xFFFF) return fail(HHV_E_LIMIT, "hhv_hit_paths_packed: query of %d rows (16-bit path records)", c->Lq);
  HIP_TRY(hipSetDevice(c->par.device));
  const int n = ts->n;
  // offsets of the compact records from the step counts the caller already holds (hhv_hits): entry 0 .. nsteps per hit
  ts->pk_off.resize((size_t)n + 1);
  ts->pk_off[0] = 0;
  for (int k = 0; k < n; ++k) {
    const int64_t cap = ts->path_off[k + 1] - ts->path_off[k];
    if (hits[k].nsteps < 0 || hits[k].nsteps + 1 > cap) return fail(HHV_E_ARG, "hhv_hit_paths_packed: hits[%d].nsteps = %d is not this set's", k, hits[k].nsteps);
    ts->pk_off[(size_t)k + 1] = ts->pk_off[k] + hits[k].nsteps + 1;
  }
  const size_t total = (size_t)ts->pk_off[n], pad = (total + 63) & ~(size_t)63;
  // one block: [offsets n+1 x 8][i: u16][j: u16][S: f32][states: i8], every piece 256-byte aligned
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_off = 0, o_i = up((size_t)(n + 1) * 8), o_j = o_i + up(pad * 2), o_S = o_j + up(pad * 2), o_s = o_S + up(pad * 4),
               bytes = o_s + up(pad);
  if (c->d_packed_bytes < bytes) {
    dfree(c->d_packed);
    c->d_packed = nullptr;
    c->d_packed_bytes = 0;
    HIP_TRY(hipMalloc(&c->d_packed, bytes + bytes / 4));
    c->d_packed_bytes = bytes + bytes / 4;
  }
  if (c->h_packed_bytes < bytes) {
    if (c->h_packed) (void)hipHostFree(c->h_packed);
    c->h_packed = nullptr;
    c->h_packed_bytes = 0;
    HIP_TRY(hipHostMalloc(&c->h_packed, bytes + bytes / 4, hipHostMallocDefault));
    c->h_packed_bytes = bytes + bytes / 4;
  }
  char* d = (char*)c->d_packed;
  char* h = (char*)c->h_packed;
  memcpy(h + o_off, ts->pk_off.data(), (size_t)(n + 1) * 8);
  HIP_TRY(hipMemcpyAsync(d + o_off, h + o_off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, c->stream));
  const int lr = launch_pack_paths(ts->d_hits, n, ts->d_path_off, ts->d_i_steps, ts->d_j_steps, ts->d_states, ts->d_S, (const int64_t*)(d + o_off),
                                   (uint16_t*)(d + o_i), (uint16_t*)(d + o_j), (int8_t*)(d + o_s), (float*)(d + o_S), c->stream);
  if (lr != 0) return fail(HHV_E_DEVICE, "hhv_hit_paths_packed: launch failed: %s", hipGetErrorString((hipError_t)(-lr)));
  HIP_TRY(hipMemcpyAsync(h + o_i, d + o_i, bytes - o_i, hipMemcpyDeviceToHost, c->stream));  // the four arrays in one copy
  {
    const int sc = sync_check(c, "hhv_hit_paths_packed");
    if (sc != HHV_OK) return sc;
  }
  ts->packed_valid = true;
  *off = ts->pk_off.data();
  *i_steps = (const uint16_t*)(h + o_i);
  *j_steps = (const uint16_t*)(h + o_j);
  *states = (const int8_t*)(h + o_s);
  *S = (const float*)(h + o_S);
  return HHV_OK;
}

int hhv_topk(hhv_ctx* c, hhv_tset* ts, int32_t k, uint32_t flags, hhv_hit* out, void* d_out, int32_t* n_out) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_topk: null argument");
  if (k < 1) return fail(HHV_E_ARG, "hhv_topk: k = %d", k);
  const bool raw = (flags & HHV_TOPK_RAW) != 0, pval = (flags & HHV_TOPK_PVALUE) != 0;
  if (!raw && !ts->hits_valid) return fail(HHV_E_STATE, "hhv_topk: call hhv_hits first (or pass HHV_TOPK_RAW)");
  if (pval && (raw || !ts->d_neff)) return fail(HHV_E_STATE, "hhv_topk: HHV_TOPK_PVALUE needs hhv_hits and hhv_tset_set_neff (and not HHV_TOPK_RAW)");
  HIP_TRY(hipSetDevice(c->par.device));
  const int kk = std::min(k, ts->n);
  if (ts->topk_cap < k) {
    tfree(ts->ctx, ts->d_topk);
    HIP_TRY(tmalloc(ts->ctx, &ts->d_topk, (size_t)k * sizeof(DevHit)));
    ts->topk_cap = k;
  }
  DevHit* dst = d_out ? (DevHit*)d_out : ts->d_topk;
  std::string err;
  if (pval) {
    if (!ts->d_rank) HIP_TRY(tmalloc(ts->ctx, &ts->d_rank, (size_t)std::max(ts->n, 1) * sizeof(float)));
    topk_rank_pvalue(ts->d_hits, ts->n, ts->d_L, ts->d_neff, c->Lq, ts->q_neff, c->par.local, ts->d_rank, c->stream);
  }
  // a small set: one launch, raw results read in place (no hit records built for the templates that are not selected)
  const int small = topk_small_device(ts->d_hits, raw ? ts->d_results : nullptr, ts->n, kk, ts->d_gids, dst, c->stream, &err,
                                      pval ? ts->d_rank : nullptr);
  if (small < 0) return fail(HHV_E_DEVICE, "hhv_topk: %s", err.c_str());
  if (small > 0) {
    if (!ts->d_keys) {
      HIP_TRY(tmalloc(ts->ctx, &ts->d_keys, (size_t)ts->n * sizeof(uint64_t)));
      HIP_TRY(tmalloc(ts->ctx, &ts->d_sorted, (size_t)ts->n * sizeof(uint64_t)));
      ts->sort_temp_bytes = topk_temp_bytes(ts->n);
      HIP_TRY(tmalloc(ts->ctx, &ts->d_sort_temp, ts->sort_temp_bytes));
    }
    const DevHit* src = ts->d_hits;
    // score-only results: read in place by the selection (K <= 1024); only the full sort of a larger K works on hit records
    const bool in_place = raw && kk <= 1024 && topk_small_enabled();
    if (raw && !in_place) {
      if (!ts->d_raw_hits) HIP_TRY(tmalloc(ts->ctx, &ts->d_raw_hits, (size_t)ts->n * sizeof(DevHit)));
      results_to_hits(ts->d_results, ts->n, ts->d_raw_hits, c->stream);
      src = ts->d_raw_hits;
    }
    if (topk_device(src, ts->n, kk, ts->d_gids, dst, ts->d_keys, ts->d_sorted, ts->d_sort_temp, ts->sort_temp_bytes, c->stream,
                    &err, pval ? ts->d_rank : nullptr, in_place ? ts->d_results : nullptr) != 0)
      return fail(HHV_E_DEVICE, "hhv_topk: %s", err.c_str());
  }
  if (kk < k) HIP_TRY(hipMemsetAsync(dst + kk, 0xFF, (size_t)(k - kk) * sizeof(DevHit), c->stream));
  if (out) {
    HIP_TRY(hipMemcpyAsync(out, dst, (size_t)k * sizeof(DevHit), hipMemcpyDeviceToHost, c->stream));
    {
      const int src_ = sync_check(c, "hhv_topk");
      if (src_ != HHV_OK) return src_;
    }
  }
  if (n_out) *n_out = kk;  // (known without the device: nothing to wait for when the records stay on the device)
  return HHV_OK;
}

int hhv_tset_set_neff(hhv_ctx* c, hhv_tset* ts, float q_neff, const float* t_neff) {
  if (!c || !ts || !t_neff) return fail(HHV_E_ARG, "hhv_tset_set_neff: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_tset_set_neff: template set belongs to another context");
  HIP_TRY(hipSetDevice(c->par.device));
  if (!ts->d_neff) HIP_TRY(tmalloc(ts->ctx, &ts->d_neff, (size_t)std::max(ts->n, 1) * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(ts->d_neff, t_neff, (size_t)ts->n * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  ts->q_neff = q_neff;
  return HHV_OK;
}

int hhv_tset_set_global_ids(hhv_ctx* c, hhv_tset* ts, const int32_t* ids) {
  if (!c || !ts) return fail(HHV_E_ARG, "hhv_tset_set_global_ids: null argument");
  if (ts->ctx != c) return fail(HHV_E_ARG, "hhv_tset_set_global_ids: template set belongs to another context");
  HIP_TRY(hipSetDevice(c->par.device));
  if (!ids) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    tfree(ts->ctx, ts->d_gids);
    ts->d_gids = nullptr;
    return HHV_OK;
  }
  for (int k = 0; k < ts->n; ++k)
    if (ids[k] < 0) return fail(HHV_E_ARG, "hhv_tset_set_global_ids: ids[%d] = %d (must be >= 0)", k, ids[k]);
  if (!ts->d_gids) HIP_TRY(tmalloc(ts->ctx, &ts->d_gids, (size_t)std::max(ts->n, 1) * sizeof(int32_t)));
  HIP_TRY(hipMemcpyAsync(ts->d_gids, ids, (size_t)ts->n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return HHV_OK;
}

int hhv_merge_hits(hhv_ctx* c, const void* d_in, int32_t m, int32_t k, hhv_hit* out, void* d_out, int32_t* n_out) {
  if (!c || !d_in) return fail(HHV_E_ARG, "hhv_merge_hits: null argument");
  if (m < 1 || k < 1) return fail(HHV_E_ARG, "hhv_merge_hits: m = %d, k = %d", m, k);
  HIP_TRY(hipSetDevice(c->par.device));
  if (c->merge_cap < k) {
    dfree(c->d_merge);
    c->d_merge = nullptr;
    c->merge_cap = 0;
    HIP_TRY(hipMalloc(&c->d_merge, (size_t)k * sizeof(DevHit) + sizeof(int)));
    c->merge_cap = k;
  }
  int* d_n = reinterpret_cast<int*>(reinterpret_cast<char*>(c->d_merge) + (size_t)c->merge_cap * sizeof(DevHit));
  DevHit* dst = d_out ? (DevHit*)d_out : (DevHit*)c->d_merge;
  std::string err;
  if (merge_hits_device((const DevHit*)d_in, m, k, dst, d_n, c->stream, &err) != 0)
    return fail(HHV_E_DEVICE, "hhv_merge_hits: %s", err.c_str());
  if (!out && !n_out) return HHV_OK;  // records and count stay on the device: asynchronous on the context's stream
  int nv = 0;
  if (out) HIP_TRY(hipMemcpyAsync(out, dst, (size_t)k * sizeof(DevHit), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&nv, d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  {
    const int src_ = sync_check(c, "hhv_merge_hits");
    if (src_ != HHV_OK) return src_;
  }
  if (n_out) *n_out = nv;
  return HHV_OK;
}

}  // extern "C"
