// rccl_runner.cpp -- see rccl_runner.h.  Talks to the GPU through the C ABI (include/hhviterbi_hip.h), to the other ranks
// through librccl; no DP arithmetic here.
#include "rccl_runner.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

namespace hhv {

namespace {
void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) throw Error(HHV_E_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
}
void nccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess) throw Error(HHV_E_DEVICE, std::string(what) + ": " + ncclGetErrorString(r));
}
void hhv_ok(int rc, const char* what) {
  if (rc != HHV_OK) throw Error(rc, std::string(what) + ": " + hhv_last_error());
}
}  // namespace

static_assert(sizeof(ncclUniqueId) == RcclShardedRunner::kIdBytes, "ncclUniqueId");

void RcclShardedRunner::MakeId(void* id) {
  ncclUniqueId u;
  nccl_ok(ncclGetUniqueId(&u), "ncclGetUniqueId");
  memcpy(id, &u, sizeof(u));
}

RcclShardedRunner::RcclShardedRunner(int world, int rank, const void* id, int device, const hhv_params& par)
    : world_(world), rank_(rank), device_(device) {
  if (world < 1 || rank < 0 || rank >= world || !id) throw Error(HHV_E_ARG, "RcclShardedRunner: bad world / rank / id");
  hip_ok(hipSetDevice(device), "hipSetDevice");
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  ncclComm_t comm;
  nccl_ok(ncclCommInitRank(&comm, world, u, rank), "ncclCommInitRank");
  comm_ = comm;
  // from here on a failure must give back what exists already: the destructor does not run for a half-built object, and a
  // communicator that is never destroyed leaves the other ranks of its (collective) initialisation hanging
  try {
    hhv_params p = par;
    p.device = device;
    hhv_ok(hhv_create(&ctx_, &p), "hhv_create");
    for (int k = 0; k < 4; ++k) {
      hipEvent_t e;
      hip_ok(hipEventCreate(&e), "hipEventCreate");
      ev_[k] = e;
    }
  } catch (...) {
    for (void*& e : ev_)
      if (e) {
        (void)hipEventDestroy((hipEvent_t)e);
        e = nullptr;
      }
    if (ctx_) hhv_destroy(ctx_);
    ctx_ = nullptr;
    (void)ncclCommAbort((ncclComm_t)comm_);
    comm_ = nullptr;
    throw;
  }
}

RcclShardedRunner::~RcclShardedRunner() {
  (void)hipSetDevice(device_);
  if (ts_) hhv_tset_free(ts_);
  if (ctx_) hhv_destroy(ctx_);
  if (d_send_) (void)hipFree(d_send_);
  if (d_recv_) (void)hipFree(d_recv_);
  for (void* e : ev_)
    if (e) (void)hipEventDestroy((hipEvent_t)e);
  if (comm_) (void)ncclCommDestroy((ncclComm_t)comm_);
}

std::vector<int32_t> RcclShardedRunner::Plan(const std::vector<int32_t>& L) const {
  std::vector<int32_t> shard_of(L.size()), mine;
  hhv_ok(hhv_shard_plan((int32_t)L.size(), L.data(), world_, shard_of.data()), "hhv_shard_plan");
  for (size_t g = 0; g < L.size(); ++g)
    if (shard_of[g] == rank_) mine.push_back((int32_t)g);
  return mine;
}

void RcclShardedRunner::Upload(const std::vector<int32_t>& ids, const std::vector<int32_t>& L, const float* const* p, const float* const* tr) {
  if (ids.size() != L.size()) throw Error(HHV_E_ARG, "RcclShardedRunner::Upload: ids / L sizes differ");
  hip_ok(hipSetDevice(device_), "hipSetDevice");
  if (ts_) hhv_tset_free(ts_);
  ts_ = nullptr;
  hhv_ok(hhv_upload_templates(ctx_, (int32_t)ids.size(), L.data(), p, tr, &ts_), "hhv_upload_templates");
  hhv_ok(hhv_tset_set_global_ids(ctx_, ts_, ids.data()), "hhv_tset_set_global_ids");
}

void RcclShardedRunner::Adopt(const std::vector<int32_t>& ids, const std::vector<int32_t>& L, const void* d_stream) {
  if (ids.size() != L.size()) throw Error(HHV_E_ARG, "RcclShardedRunner::Adopt: ids / L sizes differ");
  hip_ok(hipSetDevice(device_), "hipSetDevice");
  if (ts_) hhv_tset_free(ts_);
  ts_ = nullptr;
  hhv_ok(hhv_adopt_device_stream(ctx_, (int32_t)ids.size(), L.data(), d_stream, &ts_), "hhv_adopt_device_stream");
  hhv_ok(hhv_tset_set_global_ids(ctx_, ts_, ids.data()), "hhv_tset_set_global_ids");
}

void RcclShardedRunner::grow(int K) {
  if (K <= cap_k_) return;
  if (d_send_) (void)hipFree(d_send_);
  if (d_recv_) (void)hipFree(d_recv_);
  d_send_ = d_recv_ = nullptr;
  cap_k_ = 0;
  hip_ok(hipMalloc(&d_send_, (size_t)K * sizeof(hhv_hit)), "hipMalloc");
  hip_ok(hipMalloc(&d_recv_, (size_t)world_ * K * sizeof(hhv_hit)), "hipMalloc");
  hip_ok(hipMemset(d_send_, 0xFF, (size_t)K * sizeof(hhv_hit)), "hipMemset");  // (index -1: "no hit" for the merge)
  cap_k_ = K;
}

void RcclShardedRunner::Search(const float* q_p, const float* q_tr, int Lq, int K, bool backtrace, std::vector<hhv_hit>* out) {
  if (!ts_) throw Error(HHV_E_STATE, "RcclShardedRunner::Search: no templates uploaded");
  if (K < 1) throw Error(HHV_E_ARG, "RcclShardedRunner::Search: K < 1");
  hip_ok(hipSetDevice(device_), "hipSetDevice");
  grow(K);
  hipStream_t st = (hipStream_t)hhv_stream(ctx_);
  hip_ok(hipEventRecord((hipEvent_t)ev_[0], st), "hipEventRecord");
  hhv_ok(hhv_set_query(ctx_, q_p, q_tr, Lq), "hhv_set_query");
  hhv_ok(hhv_align_async(ctx_, ts_, backtrace ? HHV_ALIGN_BACKTRACE : 0u, nullptr), "hhv_align_async");
  if (backtrace) hhv_ok(hhv_hits(ctx_, ts_, nullptr), "hhv_hits");
  hhv_ok(hhv_topk(ctx_, ts_, K, backtrace ? 0u : HHV_TOPK_RAW, nullptr, d_send_, nullptr), "hhv_topk");
  hip_ok(hipEventRecord((hipEvent_t)ev_[1], st), "hipEventRecord");
  // the one exchange of a search: K records per rank, on the context's stream (ordered behind the top-K, in front of the merge)
  nccl_ok(ncclAllGather(d_send_, d_recv_, (size_t)K * sizeof(hhv_hit), ncclChar, (ncclComm_t)comm_, st), "ncclAllGather");
  hip_ok(hipEventRecord((hipEvent_t)ev_[2], st), "hipEventRecord");
  last_k_ = K;
  timed_ = true;
  if (out) {
    out->assign((size_t)K, hhv_hit());
    int32_t n = 0;
    hhv_ok(hhv_merge_hits(ctx_, d_recv_, world_ * K, K, out->data(), nullptr, &n), "hhv_merge_hits");  // (waits)
    out->resize((size_t)n);
    hip_ok(hipEventRecord((hipEvent_t)ev_[3], st), "hipEventRecord");
  } else {
    hhv_ok(hhv_merge_hits(ctx_, d_recv_, world_ * K, K, nullptr, nullptr, nullptr), "hhv_merge_hits");
    hip_ok(hipEventRecord((hipEvent_t)ev_[3], st), "hipEventRecord");
  }
}

void RcclShardedRunner::Wait() {
  hip_ok(hipSetDevice(device_), "hipSetDevice");
  hhv_ok(hhv_sync(ctx_), "hhv_sync");
}

RcclShardedRunner::Timing RcclShardedRunner::timing() {
  Timing t;
  if (!timed_) return t;
  hip_ok(hipEventSynchronize((hipEvent_t)ev_[3]), "hipEventSynchronize");
  hip_ok(hipEventElapsedTime(&t.local, (hipEvent_t)ev_[0], (hipEvent_t)ev_[1]), "hipEventElapsedTime");
  hip_ok(hipEventElapsedTime(&t.all_gather, (hipEvent_t)ev_[1], (hipEvent_t)ev_[2]), "hipEventElapsedTime");
  hip_ok(hipEventElapsedTime(&t.merge, (hipEvent_t)ev_[2], (hipEvent_t)ev_[3]), "hipEventElapsedTime");
  hhv_ok(hhv_last_kernel_ms(ctx_, &t.dp_kernel), "hhv_last_kernel_ms");
  return t;
}

int64_t RcclShardedRunner::cells(int Lq) const { return ts_ ? hhv_tset_cells(ts_, Lq) : 0; }

void RcclShardedRunner::AllGatherHost(const void* mine, void* all, size_t bytes) {
  hip_ok(hipSetDevice(device_), "hipSetDevice");
  hipStream_t st = (hipStream_t)hhv_stream(ctx_);
  void *d_in = nullptr, *d_out = nullptr;
  hip_ok(hipMalloc(&d_in, bytes), "hipMalloc");
  hip_ok(hipMalloc(&d_out, bytes * (size_t)world_), "hipMalloc");
  hip_ok(hipMemcpyAsync(d_in, mine, bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync");
  nccl_ok(ncclAllGather(d_in, d_out, bytes, ncclChar, (ncclComm_t)comm_, st), "ncclAllGather");
  hip_ok(hipMemcpyAsync(all, d_out, bytes * (size_t)world_, hipMemcpyDeviceToHost, st), "hipMemcpyAsync");
  hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize");
  (void)hipFree(d_in);
  (void)hipFree(d_out);
}

void RcclShardedRunner::Barrier() {
  char one = 0;
  std::vector<char> all((size_t)world_);
  AllGatherHost(&one, all.data(), 1);
}

}  // namespace hhv
