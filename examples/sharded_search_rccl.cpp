// examples/sharded_search_rccl.cpp -- the sharded search of SURVEY.md 8(e) as a native multi-process program: one process
// per GPU, the C ABI (include/hhviterbi_hip.h) + librccl directly, no Python, no torch.
//
//   hhv_shard_plan  ->  per rank: hhv_upload_templates(own shard) + hhv_tset_set_global_ids
//   per query:          hhv_set_query -> hhv_align_async -> [hhv_hits] -> hhv_topk(d_send)      (all on hhv_stream(ctx))
//                       ncclAllGather(d_send -> d_recv, K records per rank)                      (on the same stream)
//                       hhv_merge_hits(d_recv, world x K) -> the K best, identical on every rank
//
// This is what HH-suite's own multi-process driver would become: hhblits_mpi splits the database over ranks and merges the
// hit lists on the master (src/hhblits_mpi.cpp:135-231); inside one process the reference appends the batches' hits serially
// (src/hhviterbirunner.cpp:117-122,173).  Here whole templates are distributed, no DP data crosses GPUs, and the only
// exchange is ONE all-gather of K x 40 bytes per rank over xGMI.
//
// Launch:  sharded_search_rccl --world N [--templates n] [--lq L] [--lt L] [--topk K] [--backtrace] [--zipf] [--steps S] [--check]
//   the launcher forks N ranks of itself (rank r sees GPU r through HIP_VISIBLE_DEVICES when at least N GPUs are visible),
//   rank 0 creates the ncclUniqueId and hands it to the others through a file; no MPI needed.
//   --check: rank 0 also aligns the WHOLE database on its own GPU and compares the merged list with that search's top K.
// Build:   make example_rccl     (hipcc, -lrccl -lhhviterbi_hip)
#include <hip/hip_runtime.h>
#include <math.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../include/hhviterbi_hip.h"

namespace {

#define HIP_OK(x)                                                                          \
  do {                                                                                     \
    hipError_t e_ = (x);                                                                   \
    if (e_ != hipSuccess) {                                                                \
      fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #x, hipGetErrorString(e_));            \
      exit(3);                                                                             \
    }                                                                                      \
  } while (0)
#define NCCL_OK(x)                                                                         \
  do {                                                                                     \
    ncclResult_t r_ = (x);                                                                 \
    if (r_ != ncclSuccess) {                                                               \
      fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #x, ncclGetErrorString(r_));           \
      exit(3);                                                                             \
    }                                                                                      \
  } while (0)
#define HHV_OK_(x)                                                                         \
  do {                                                                                     \
    int r_ = (x);                                                                          \
    if (r_ != HHV_OK) {                                                                    \
      fprintf(stderr, "[rank %d] %s: %d %s\n", g_rank, #x, r_, hhv_last_error());          \
      exit(3);                                                                             \
    }                                                                                      \
  } while (0)

int g_rank = 0;

struct Rng {  // splitmix64
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double u() { return (double)(next() >> 11) / 9007199254740992.0; }
};
const float kPb[20] = {0.0787f, 0.0512f, 0.0448f, 0.0536f, 0.0135f, 0.0403f, 0.0610f, 0.0688f, 0.0229f, 0.0590f,
                       0.0964f, 0.0593f, 0.0237f, 0.0396f, 0.0483f, 0.0683f, 0.0585f, 0.0132f, 0.0321f, 0.0668f};

// a prepared profile: p[(L+1)*20] (odds for templates), tr[(L+1)*7] log2 transitions (src/hhdecl.h:68 order).  A template's
// content depends on its GLOBAL id alone, so every rank can materialise exactly its own shard of one global database.
struct Profile {
  int L;
  std::vector<float> p, tr;
};
Profile make_profile(uint64_t seed, int L, bool is_query, const Profile* homolog_of) {
  Rng rng(seed);
  Profile h;
  h.L = L;
  h.p.assign((size_t)(L + 1) * 20, 0.f);
  h.tr.assign((size_t)(L + 1) * 7, 0.f);
  for (int i = 1; i <= L; ++i) {
    double col[20], sum = 0;
    for (int a = 0; a < 20; ++a) sum += (col[a] = pow(rng.u(), 6.0) + 1e-9);
    for (int a = 0; a < 20; ++a) {
      double v = 0.7 * col[a] / sum + 0.3 * kPb[a];
      if (homolog_of && i <= homolog_of->L) v = 0.75 * homolog_of->p[(size_t)i * 20 + a] + 0.25 * v;
      h.p[(size_t)i * 20 + a] = is_query ? (float)v : (float)(v / kPb[a]);
    }
  }
  for (int i = 0; i <= L; ++i) {
    const double pI = 0.01 + 0.04 * rng.u(), pD = 0.01 + 0.04 * rng.u();
    float* t = &h.tr[(size_t)i * 7];
    t[0] = (float)log2(1 - pI - pD);
    t[1] = (float)(0.6 * log2(pI));
    t[2] = (float)(0.6 * log2(pD));
    t[3] = (float)log2(0.6);
    t[4] = (float)(0.6 * log2(0.4));
    t[5] = (float)log2(0.6);
    t[6] = (float)(0.6 * log2(0.4));
    if (i == 0 || i == L) {
      t[0] = 0.f;
      t[1] = t[2] = -100000.f;
    }
    if (i == L) {
      t[6] = -100000.f;
      t[5] = 0.f;
    }
  }
  return h;
}

struct Options {
  int world = 1, rank = -1, n = 20000, lq = 300, lt = 300, topk = 500, steps = 5, backtrace = 0, zipf = 0, check = 0;
  std::string id_file;
};

int zipf_length(uint64_t id) {  // BASELINE configs[4]: 49 + k, k ~ Zipf(1.2) truncated to 1..951
  static std::vector<double> cdf;
  if (cdf.empty()) {
    double s = 0;
    for (int k = 1; k <= 951; ++k) cdf.push_back(s += pow((double)k, -1.2));
    for (double& c : cdf) c /= s;
  }
  Rng r(0x21FF0000ull + id);
  const double u = r.u();
  return 49 + 1 + (int)(std::lower_bound(cdf.begin(), cdf.end(), u) - cdf.begin());
}

struct Shard {
  hhv_ctx* ctx = nullptr;
  hhv_tset* ts = nullptr;
  int n = 0;
  int64_t cells = 0;
};

// upload the templates `ids` (global ids) of the global database as one resident set
Shard make_shard(const Options& o, const Profile& q, const std::vector<int32_t>& L, const std::vector<int32_t>& ids) {
  Shard s;
  hhv_params par;
  memset(&par, 0, sizeof(par));
  par.device = 0;
  par.local = o.zipf ? 1 : 0;
  par.shift = -0.03f;
  par.corr = 0.1f;
  par.ssw = 0.11f;
  par.ss_mode = 2;
  HHV_OK_(hhv_create(&s.ctx, &par));
  std::vector<Profile> prof;
  prof.reserve(ids.size());
  std::vector<const float*> pp, tt;
  std::vector<int32_t> Ls;
  for (int32_t id : ids) {
    prof.push_back(make_profile(0x5EED0000ull + (uint64_t)id, L[id], false, id % 50 == 0 ? &q : nullptr));
    Ls.push_back(L[id]);
    s.cells += (int64_t)o.lq * L[id];
  }
  for (const Profile& p : prof) {
    pp.push_back(p.p.data());
    tt.push_back(p.tr.data());
  }
  s.n = (int)ids.size();
  HHV_OK_(hhv_upload_templates(s.ctx, s.n, Ls.data(), pp.data(), tt.data(), &s.ts));
  HHV_OK_(hhv_tset_set_global_ids(s.ctx, s.ts, ids.data()));
  return s;
}

// one search on a shard up to its local top K in d_send (device, K records); nothing waits for the device
void search_local(const Options& o, Shard& s, const Profile& q, void* d_send) {
  HHV_OK_(hhv_set_query(s.ctx, q.p.data(), q.tr.data(), q.L));
  HHV_OK_(hhv_align_async(s.ctx, s.ts, o.backtrace ? HHV_ALIGN_BACKTRACE : 0u, nullptr));
  if (o.backtrace) HHV_OK_(hhv_hits(s.ctx, s.ts, nullptr));
  HHV_OK_(hhv_topk(s.ctx, s.ts, o.topk, o.backtrace ? 0u : HHV_TOPK_RAW, nullptr, d_send, nullptr));
}

int run_rank(const Options& o) {
  g_rank = o.rank;
  // ---- communicator: rank 0 creates the id, the others read it from the file
  ncclUniqueId id;
  if (o.rank == 0) {
    NCCL_OK(ncclGetUniqueId(&id));
    const std::string tmp = o.id_file + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(&id, sizeof(id), 1, f) != 1) return 4;
    fclose(f);
    rename(tmp.c_str(), o.id_file.c_str());
  } else {
    for (int tries = 0;; ++tries) {
      FILE* f = fopen(o.id_file.c_str(), "rb");
      if (f) {
        const size_t got = fread(&id, sizeof(id), 1, f);
        fclose(f);
        if (got == 1) break;
      }
      if (tries > 6000) {
        fprintf(stderr, "[rank %d] no id file\n", o.rank);
        return 4;
      }
      usleep(10000);
    }
  }
  HIP_OK(hipSetDevice(0));  // (the launcher narrowed HIP_VISIBLE_DEVICES to this rank's GPU)
  ncclComm_t comm;
  NCCL_OK(ncclCommInitRank(&comm, o.world, id, o.rank));

  // ---- the global database: lengths and the plan are computed by every rank, profiles only for the own shard
  std::vector<int32_t> L(o.n);
  for (int k = 0; k < o.n; ++k) L[k] = o.zipf ? zipf_length((uint64_t)k) : o.lt;
  std::vector<int32_t> shard_of(o.n);
  HHV_OK_(hhv_shard_plan(o.n, L.data(), o.world, shard_of.data()));
  std::vector<int32_t> mine;
  for (int k = 0; k < o.n; ++k)
    if (shard_of[k] == o.rank) mine.push_back(k);
  const Profile q = make_profile(0x51000000ull, o.lq, true, nullptr);
  Shard s = make_shard(o, q, L, mine);

  const size_t rec = sizeof(hhv_hit);
  void *d_send = nullptr, *d_recv = nullptr;
  HIP_OK(hipMalloc(&d_send, (size_t)o.topk * rec));
  HIP_OK(hipMalloc(&d_recv, (size_t)o.world * o.topk * rec));
  hipStream_t stream = (hipStream_t)hhv_stream(s.ctx);
  hipEvent_t e0, e1, e2, e3;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  HIP_OK(hipEventCreate(&e2));
  HIP_OK(hipEventCreate(&e3));
  std::vector<hhv_hit> merged(o.topk);
  int32_t n_merged = 0;

  double best_wall = 1e30, sum_ag = 0, sum_kernel = 0, sum_local = 0;
  for (int step = -1; step < o.steps; ++step) {  // step -1 = warm-up (communicator set-up, first launches)
    HIP_OK(hipStreamSynchronize(stream));
    NCCL_OK(ncclAllGather(d_send, d_recv, 1, ncclChar, comm, stream));  // (a barrier across the ranks)
    HIP_OK(hipStreamSynchronize(stream));
    const auto t0 = std::chrono::steady_clock::now();
    HIP_OK(hipEventRecord(e0, stream));
    search_local(o, s, q, d_send);
    HIP_OK(hipEventRecord(e1, stream));
    NCCL_OK(ncclAllGather(d_send, d_recv, (size_t)o.topk * rec, ncclChar, comm, stream));
    HIP_OK(hipEventRecord(e2, stream));
    HHV_OK_(hhv_merge_hits(s.ctx, d_recv, o.world * o.topk, o.topk, nullptr, nullptr, nullptr));
    HIP_OK(hipEventRecord(e3, stream));
    HIP_OK(hipStreamSynchronize(stream));
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    float ms_local = 0, ms_ag = 0, ms_kernel = 0;
    HIP_OK(hipEventElapsedTime(&ms_local, e0, e1));
    HIP_OK(hipEventElapsedTime(&ms_ag, e1, e2));
    HHV_OK_(hhv_last_kernel_ms(s.ctx, &ms_kernel));
    if (step >= 0) {
      best_wall = std::min(best_wall, wall);
      sum_ag += ms_ag;
      sum_kernel += ms_kernel;
      sum_local += ms_local;
    }
  }
  // the merged list on the host (same on every rank)
  HHV_OK_(hhv_merge_hits(s.ctx, d_recv, o.world * o.topk, o.topk, merged.data(), nullptr, &n_merged));
  printf("rank %d/%d: shard %d templates, %.3e cells, DP kernel %.3f ms, local step %.3f ms, all-gather (incl. wait for the slowest rank) %.3f ms, "
         "best step %.3f ms -> %.3e cells/s on this rank\n",
         o.rank, o.world, s.n, (double)s.cells, sum_kernel / o.steps, sum_local / o.steps, sum_ag / o.steps, best_wall * 1e3,
         (double)s.cells / best_wall);

  int rc = 0;
  // every rank must hold the same merged list: compare checksums through one more all-gather
  {
    uint64_t h = 1469598103934665603ull;
    for (int t = 0; t < n_merged; ++t) {
      const unsigned char* b = reinterpret_cast<const unsigned char*>(&merged[t]);
      for (size_t x = 0; x < rec; ++x) h = (h ^ b[x]) * 1099511628211ull;
    }
    uint64_t *d_h = nullptr, *d_all = nullptr;
    HIP_OK(hipMalloc(&d_h, 8));
    HIP_OK(hipMalloc(&d_all, 8 * (size_t)o.world));
    HIP_OK(hipMemcpy(d_h, &h, 8, hipMemcpyHostToDevice));
    NCCL_OK(ncclAllGather(d_h, d_all, 8, ncclChar, comm, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::vector<uint64_t> all(o.world);
    HIP_OK(hipMemcpy(all.data(), d_all, 8 * (size_t)o.world, hipMemcpyDeviceToHost));
    for (int r = 0; r < o.world; ++r)
      if (all[r] != h) rc = 5;
    if (o.rank == 0) printf("merged list: %d hits, checksum %016llx, %s on all %d ranks\n", n_merged, (unsigned long long)h,
                            rc == 0 ? "identical" : "DIFFERENT", o.world);
    (void)hipFree(d_h);
    (void)hipFree(d_all);
  }
  if (o.check && o.rank == 0) {
    // the same database on ONE GPU: its top K must be the merged list, record for record
    std::vector<int32_t> all_ids(o.n);
    for (int k = 0; k < o.n; ++k) all_ids[k] = k;
    Shard whole = make_shard(o, q, L, all_ids);
    std::vector<hhv_hit> ref(o.topk);
    int32_t n_ref = 0;
    HHV_OK_(hhv_set_query(whole.ctx, q.p.data(), q.tr.data(), q.L));
    HHV_OK_(hhv_align(whole.ctx, whole.ts, o.backtrace ? HHV_ALIGN_BACKTRACE : 0u, nullptr));
    if (o.backtrace) HHV_OK_(hhv_hits(whole.ctx, whole.ts, nullptr));
    HHV_OK_(hhv_topk(whole.ctx, whole.ts, o.topk, o.backtrace ? 0u : HHV_TOPK_RAW, ref.data(), nullptr, &n_ref));
    const bool same = n_ref == n_merged && memcmp(ref.data(), merged.data(), (size_t)n_ref * rec) == 0;
    printf("check against ONE GPU holding all %d templates: %s (best hit: template %d, score %.4f)\n", o.n,
           same ? "OK, identical records" : "MISMATCH", merged[0].index, merged[0].score);
    if (!same) rc = 6;
    hhv_tset_free(whole.ts);
    hhv_destroy(whole.ctx);
  }
  hhv_tset_free(s.ts);
  hhv_destroy(s.ctx);
  (void)hipFree(d_send);
  (void)hipFree(d_recv);
  NCCL_OK(ncclCommDestroy(comm));
  return rc;
}

}  // namespace

int main(int argc, char** argv) {
  Options o;
  for (int a = 1; a < argc; ++a) {
    const std::string k = argv[a];
    auto val = [&]() { return a + 1 < argc ? atoi(argv[++a]) : 0; };
    if (k == "--world") o.world = val();
    else if (k == "--rank") o.rank = val();
    else if (k == "--templates") o.n = val();
    else if (k == "--lq") o.lq = val();
    else if (k == "--lt") o.lt = val();
    else if (k == "--topk") o.topk = val();
    else if (k == "--steps") o.steps = val();
    else if (k == "--backtrace") o.backtrace = 1;
    else if (k == "--zipf") o.zipf = 1;
    else if (k == "--check") o.check = 1;
    else if (k == "--id-file" && a + 1 < argc) o.id_file = argv[++a];
    else {
      fprintf(stderr, "unknown option %s\n", k.c_str());
      return 2;
    }
  }
  if (o.world < 1 || o.n < o.world || o.topk < 1 || o.steps < 1) return 2;
  if (o.rank >= 0) return run_rank(o);

  // ---- launcher: fork one process per rank; rank r gets GPU r when enough GPUs are visible
  int32_t n_dev = 0;
  if (hhv_device_count(&n_dev) != HHV_OK || n_dev < 1) {
    fprintf(stderr, "no HIP device: %s\n", hhv_last_error());
    return 3;
  }
  if (o.world > n_dev) {
    fprintf(stderr, "--world %d needs %d GPUs, %d visible (RCCL does not put two ranks on one device)\n", o.world, o.world, n_dev);
    return 7;
  }
  char idf[256];
  snprintf(idf, sizeof(idf), "/tmp/hhv_rccl_id_%d", (int)getpid());
  unlink(idf);
  // the devices this process may use, as the ranks' HIP_VISIBLE_DEVICES entries
  std::vector<std::string> devs;
  if (const char* vis = getenv("HIP_VISIBLE_DEVICES")) {
    std::string v = vis, cur;
    for (char ch : v + ",") {
      if (ch == ',') {
        if (!cur.empty()) devs.push_back(cur);
        cur.clear();
      } else {
        cur.push_back(ch);
      }
    }
  }
  std::vector<pid_t> kids;
  for (int r = 0; r < o.world; ++r) {
    const pid_t pid = fork();
    if (pid == 0) {
      const std::string dev = r < (int)devs.size() ? devs[r] : std::to_string(r);
      setenv("HIP_VISIBLE_DEVICES", dev.c_str(), 1);
      std::vector<std::string> args(argv, argv + argc);
      args.push_back("--rank");
      args.push_back(std::to_string(r));
      args.push_back("--id-file");
      args.push_back(idf);
      std::vector<char*> cargs;
      for (std::string& s : args) cargs.push_back(&s[0]);
      cargs.push_back(nullptr);
      execv("/proc/self/exe", cargs.data());
      _exit(127);
    }
    kids.push_back(pid);
  }
  int rc = 0;
  for (pid_t pid : kids) {
    int st = 0;
    waitpid(pid, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = WIFEXITED(st) ? WEXITSTATUS(st) : 9;
  }
  unlink(idf);
  printf("%s\n", rc == 0 ? "sharded_search_rccl: OK" : "sharded_search_rccl: FAILED");
  return rc;
}
