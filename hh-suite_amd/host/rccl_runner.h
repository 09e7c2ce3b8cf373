// rccl_runner.h -- the template database sharded over the GPUs of a node with ONE PROCESS PER GPU: the multi-process form of
// hhv::ShardedViterbiRunner (viterbi_runner.h), for hosts that are multi-process already.  HH-suite's own multi-process driver
// splits the database over MPI ranks and merges the hit lists on the master (src/hhblits_mpi.cpp:135-231); inside one process the
// reference appends the batches' hits serially (src/hhviterbirunner.cpp:117-122,173).  Here every rank holds whole templates
// (hhv_shard_plan: the same plan on every rank from the same global length vector), aligns them on its own GPU and the ranks
// exchange ONE ncclAllGather of K hit records (40 bytes each) per search over xGMI; the merge (hhv_merge_hits: score
// descending, global template id ascending) runs on every rank, so every rank ends with the same K best hits of the WHOLE
// database.  No DP data crosses GPUs; alternative-alignment rounds stay on the owning GPU.
//
// Built as its own library (make rccl_runner -> libhhv_rccl_runner.so: links librccl; the C ABI library does not), on top of
// include/hhviterbi_hip.h only.  How the ranks come to exist (mpirun, fork, a job scheduler) and how rank 0's id reaches the
// others (a file, MPI_Bcast, a socket) is the host's business: examples/sharded_search_rccl.cpp forks and uses a file.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/hhviterbi_hip.h"
#include "viterbi_runner.h"  // hhv::Error

namespace hhv {

class RcclShardedRunner {
 public:
  static constexpr int kIdBytes = 128;  // sizeof(ncclUniqueId)
  // rank 0 makes the id of the communicator; every rank needs the same bytes before it constructs its runner
  static void MakeId(void* id /* kIdBytes */);

  // device: the GPU of this rank as THIS process numbers it (all devices stay visible: hipSetDevice(device), so that RCCL sees
  // the peers and uses xGMI peer-to-peer rather than host staging).  Collective: returns when all `world` ranks have joined.
  RcclShardedRunner(int world, int rank, const void* id, int device, const hhv_params& par);
  ~RcclShardedRunner();
  RcclShardedRunner(const RcclShardedRunner&) = delete;
  RcclShardedRunner& operator=(const RcclShardedRunner&) = delete;

  int world() const { return world_; }
  int rank() const { return rank_; }

  // The ONE global database: L[g] = columns of the template with global id g.  Returns the global ids of this rank's shard
  // (ascending).  Pure function of (L, world, rank): every rank computes the same plan.
  std::vector<int32_t> Plan(const std::vector<int32_t>& L) const;

  // this rank's templates: prepared profiles p[k][(L+1)*20], tr[k][(L+1)*7] of the templates ids[k] (what Plan returned, or any
  // other assignment all ranks agree on); replaces an earlier set
  void Upload(const std::vector<int32_t>& ids, const std::vector<int32_t>& L, const float* const* p, const float* const* tr);
  // ... or a packed record stream that already lives on this rank's device (hhv_adopt_device_stream)
  void Adopt(const std::vector<int32_t>& ids, const std::vector<int32_t>& L, const void* d_stream);

  // One search of the whole database.  Enqueues, on the context's stream: query H2D, the DP over this rank's shard, [backtrace
  // walk + Hit scores,] the shard's K best, ONE all-gather of K records per rank, the merge.  out != nullptr: waits and copies
  // the merged list (identical on every rank) to the host; out == nullptr: nothing waits for the device (Wait() does).
  void Search(const float* q_p, const float* q_tr, int Lq, int K, bool backtrace, std::vector<hhv_hit>* out);
  void Wait();

  // timings of the last Search that was waited for (milliseconds, HIP events on the context's stream)
  struct Timing {
    float dp_kernel = 0, local = 0, all_gather = 0, merge = 0;
  };
  Timing timing();

  hhv_ctx* ctx() { return ctx_; }
  hhv_tset* tset() { return ts_; }
  int64_t cells(int Lq) const;  // Lq x (columns of this rank's templates)
  // small host-side collectives on the runner's communicator (reports, checks): a barrier across the ranks, and an all-gather
  // of `bytes` per rank from / to HOST memory (staged through the device)
  void Barrier();
  void AllGatherHost(const void* mine, void* all, size_t bytes);

  // the transport RCCL reports for this communicator is in its own log (NCCL_DEBUG=INFO); tools/scale8.sh captures it

 private:
  void grow(int K);
  int world_, rank_, device_;
  void* comm_ = nullptr;      // ncclComm_t
  hhv_ctx* ctx_ = nullptr;
  hhv_tset* ts_ = nullptr;
  void* d_send_ = nullptr;    // K records
  void* d_recv_ = nullptr;    // world x K records
  int cap_k_ = 0, last_k_ = 0;
  void* ev_[4] = {nullptr, nullptr, nullptr, nullptr};
  bool timed_ = false;
};

}  // namespace hhv
