// examples/sharded_search_rccl.cpp -- the sharded search of SURVEY.md 8(e) as a native multi-process program: one process
// per GPU on top of hhv::RcclShardedRunner (hh-suite_amd/host/rccl_runner.h: the C ABI + librccl, no Python, no torch).
//
//   every rank:  RcclShardedRunner(world, rank, id, device) -> Plan(global lengths) -> Upload(own shard)
//   per query:   Search(q, K): hhv_set_query -> hhv_align_async -> [hhv_hits] -> hhv_topk -> ncclAllGather (K records per rank)
//                -> hhv_merge_hits, all on the context's stream: the K best of the WHOLE database, identical on every rank
//
// This is what HH-suite's own multi-process driver would become: hhblits_mpi splits the database over ranks and merges the
// hit lists on the master (src/hhblits_mpi.cpp:135-231); inside one process the reference appends the batches' hits serially
// (src/hhviterbirunner.cpp:117-122,173).  Here whole templates are distributed, no DP data crosses GPUs, and the only
// exchange is ONE all-gather of K x 40 bytes per rank over xGMI.  The database is SURVEY 8(d)'s (examples/synth8d.h: the
// generator of bench.py in C++), a template's columns depend on its global id alone.
//
// Launch:  sharded_search_rccl --world N [--templates n] [--lq L] [--lt L] [--topk K] [--backtrace] [--zipf] [--steps S] [--check] [--json]
//   the launcher forks N ranks of itself; ALL GPUs stay visible to every rank, rank r uses device r (RCCL then sees its peers
//   and takes the xGMI peer-to-peer transport; NCCL_DEBUG=INFO shows which), rank 0 creates the id and hands it to the others
//   through a file; no MPI needed.
//   --check: rank 0 also aligns the WHOLE database on its own GPU and compares the merged list with that search's top K.
//   --json:  rank 0 prints one JSON line (per-rank DP kernel / local / all-gather / merge times, cells/s) for tools/scale8.sh
// Build:   make example_rccl     (hipcc, -lhhv_rccl_runner -lhhviterbi_hip -lrccl)
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

#include "../hh-suite_amd/host/rccl_runner.h"
#include "../include/hhviterbi_hip.h"
#include "synth8d.h"

namespace {

struct Options {
  int world = 1, rank = -1, n = 20000, lq = 300, lt = 300, topk = 500, steps = 5, backtrace = 0, zipf = 0, check = 0, json = 0;
  std::string id_file;
};

hhv_params search_params(const Options& o, int device) {
  hhv_params par;
  memset(&par, 0, sizeof(par));
  par.device = device;
  par.local = o.zipf ? 1 : 0;   // (global mode + mixed-length SIMD batches is the reference's batch-composition quirk: SURVEY 8d config 5)
  par.shift = -0.03f;
  par.corr = 0.1f;
  par.ssw = 0.11f;
  par.ss_mode = 2;
  return par;
}

// the prepared profiles of the templates `ids` of the global database
struct Profiles {
  std::vector<synth8d::Profile> prof;
  std::vector<const float*> pp, tt;
  std::vector<int32_t> Ls;
  Profiles(const std::vector<int32_t>& L, const std::vector<int32_t>& ids) {
    prof.reserve(ids.size());
    for (int32_t id : ids) {
      prof.push_back(synth8d::make_template(id, L[id]));
      Ls.push_back(L[id]);
    }
    for (const synth8d::Profile& p : prof) {
      pp.push_back(p.p.data());
      tt.push_back(p.tr.data());
    }
  }
};

bool read_id(const std::string& path, void* id) {
  for (int tries = 0; tries < 6000; ++tries) {
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
      const size_t got = fread(id, hhv::RcclShardedRunner::kIdBytes, 1, f);
      fclose(f);
      if (got == 1) return true;
    }
    usleep(10000);
  }
  return false;
}

int run_rank(const Options& o) {
  try {
    // ---- communicator: rank 0 creates the id, the others read it from the file
    char id[hhv::RcclShardedRunner::kIdBytes];
    if (o.rank == 0) {
      hhv::RcclShardedRunner::MakeId(id);
      const std::string tmp = o.id_file + ".tmp";
      FILE* f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(id, sizeof(id), 1, f) != 1) return 4;
      fclose(f);
      rename(tmp.c_str(), o.id_file.c_str());
    } else if (!read_id(o.id_file, id)) {
      fprintf(stderr, "[rank %d] no id file\n", o.rank);
      return 4;
    }
    int32_t n_dev = 0;
    if (hhv_device_count(&n_dev) != HHV_OK) return 3;
    hhv::RcclShardedRunner runner(o.world, o.rank, id, o.rank % n_dev, search_params(o, o.rank % n_dev));

    // ---- the global database: lengths and the plan are computed by every rank, profiles only for the own shard
    std::vector<int32_t> L(o.n);
    for (int k = 0; k < o.n; ++k) L[k] = o.zipf ? synth8d::zipf_length(k) : o.lt;
    const std::vector<int32_t> mine = runner.Plan(L);
    const synth8d::Profile q = synth8d::make_query(o.lq);
    {
      Profiles shard(L, mine);
      runner.Upload(mine, shard.Ls, shard.pp.data(), shard.tt.data());
    }
    const int64_t cells = runner.cells(o.lq);

    std::vector<hhv_hit> merged;
    double best_wall = 1e30;
    hhv::RcclShardedRunner::Timing sum;
    for (int step = -1; step < o.steps; ++step) {  // step -1 = warm-up (communicator set-up, first launches)
      runner.Barrier();
      const auto t0 = std::chrono::steady_clock::now();
      runner.Search(q.p.data(), q.tr.data(), q.L, o.topk, o.backtrace != 0, nullptr);
      runner.Wait();
      const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (step >= 0) {
        const hhv::RcclShardedRunner::Timing t = runner.timing();
        best_wall = std::min(best_wall, wall);
        sum.dp_kernel += t.dp_kernel;
        sum.local += t.local;
        sum.all_gather += t.all_gather;
        sum.merge += t.merge;
      }
    }
    runner.Search(q.p.data(), q.tr.data(), q.L, o.topk, o.backtrace != 0, &merged);  // the merged list on the host (same on every rank)
    const int n_merged = (int)merged.size();
    printf("rank %d/%d: shard %d templates, %.3e cells, DP kernel %.3f ms, local step %.3f ms, all-gather (incl. wait for the slowest rank) %.3f ms, "
           "merge %.3f ms, best step %.3f ms -> %.3e cells/s on this rank\n",
           o.rank, o.world, (int)mine.size(), (double)cells, sum.dp_kernel / o.steps, sum.local / o.steps, sum.all_gather / o.steps,
           sum.merge / o.steps, best_wall * 1e3, (double)cells / best_wall);

    int rc = 0;
    // every rank must hold the same merged list, and the whole job's rate is the slowest rank's: one exchange of
    // (checksum, cells, best step, times) per rank - through the runner's own communicator
    uint64_t h = 1469598103934665603ull;
    for (int t = 0; t < n_merged; ++t) {
      const unsigned char* b = reinterpret_cast<const unsigned char*>(&merged[t]);
      for (size_t x = 0; x < sizeof(hhv_hit); ++x) h = (h ^ b[x]) * 1099511628211ull;
    }
    double mine_rec[8] = {0, (double)cells, best_wall, sum.dp_kernel / o.steps, sum.local / o.steps, sum.all_gather / o.steps, sum.merge / o.steps,
                          (double)mine.size()};
    memcpy(&mine_rec[0], &h, 8);
    std::vector<double> all((size_t)8 * o.world);
    runner.AllGatherHost(mine_rec, all.data(), sizeof(mine_rec));
    double total_cells = 0, worst = 0;
    for (int r = 0; r < o.world; ++r) {
      uint64_t hr;
      memcpy(&hr, &all[(size_t)8 * r], 8);
      if (hr != h) rc = 5;
      total_cells += all[(size_t)8 * r + 1];
      worst = std::max(worst, all[(size_t)8 * r + 2]);
    }
    if (o.rank == 0)
      printf("merged list: %d hits, checksum %016llx, %s on all %d ranks; whole job %.3e cells/s (all ranks' cells / slowest rank's best step)\n",
             n_merged, (unsigned long long)h, rc == 0 ? "identical" : "DIFFERENT", o.world, total_cells / worst);
    bool check_ok = true;
    if (o.check && o.rank == 0) {
      // the same database on ONE GPU: its top K must be the merged list, record for record
      std::vector<int32_t> all_ids(o.n);
      for (int k = 0; k < o.n; ++k) all_ids[k] = k;
      Profiles whole(L, all_ids);
      hhv_ctx* c = nullptr;
      hhv_tset* ts = nullptr;
      const hhv_params par = search_params(o, 0);
      std::vector<hhv_hit> ref(o.topk);
      int32_t n_ref = 0;
      if (hhv_create(&c, &par) != HHV_OK || hhv_upload_templates(c, o.n, whole.Ls.data(), whole.pp.data(), whole.tt.data(), &ts) != HHV_OK ||
          hhv_tset_set_global_ids(c, ts, all_ids.data()) != HHV_OK || hhv_set_query(c, q.p.data(), q.tr.data(), q.L) != HHV_OK ||
          hhv_align(c, ts, o.backtrace ? HHV_ALIGN_BACKTRACE : 0u, nullptr) != HHV_OK || (o.backtrace && hhv_hits(c, ts, nullptr) != HHV_OK) ||
          hhv_topk(c, ts, o.topk, o.backtrace ? 0u : HHV_TOPK_RAW, ref.data(), nullptr, &n_ref) != HHV_OK) {
        fprintf(stderr, "check: %s\n", hhv_last_error());
        return 6;
      }
      check_ok = n_ref == n_merged && memcmp(ref.data(), merged.data(), (size_t)n_ref * sizeof(hhv_hit)) == 0;
      printf("check against ONE GPU holding all %d templates: %s (best hit: template %d, score %.4f)\n", o.n,
             check_ok ? "OK, identical records" : "MISMATCH", n_merged ? merged[0].index : -1, n_merged ? merged[0].score : 0.f);
      if (!check_ok) rc = 6;
      hhv_tset_free(ts);
      hhv_destroy(c);
    }
    if (o.json && o.rank == 0) {
      printf("{\"program\": \"sharded_search_rccl\", \"world\": %d, \"templates\": %d, \"Lq\": %d, \"Lt\": \"%s\", \"backtrace\": %d, \"topk\": %d, "
             "\"cells_per_s\": %.6e, \"slowest_rank_best_step_ms\": %.4f, \"merged_identical_on_all_ranks\": %s, \"check_one_gpu\": %s, \"per_rank\": [",
             o.world, o.n, o.lq, o.zipf ? "zipf50-1000" : std::to_string(o.lt).c_str(), o.backtrace, o.topk, total_cells / worst, worst * 1e3,
             rc == 5 ? "false" : "true", !o.check ? "null" : check_ok ? "true" : "false");
      for (int r = 0; r < o.world; ++r)
        printf("%s{\"rank\": %d, \"templates\": %d, \"cells\": %.0f, \"best_step_ms\": %.4f, \"dp_kernel_ms\": %.4f, \"local_ms\": %.4f, \"all_gather_ms\": %.4f, \"merge_ms\": %.4f}",
               r ? ", " : "", r, (int)all[(size_t)8 * r + 7], all[(size_t)8 * r + 1], all[(size_t)8 * r + 2] * 1e3, all[(size_t)8 * r + 3],
               all[(size_t)8 * r + 4], all[(size_t)8 * r + 5], all[(size_t)8 * r + 6]);
      printf("]}\n");
    }
    return rc;
  } catch (const hhv::Error& e) {
    fprintf(stderr, "[rank %d] %s (status %d)\n", o.rank, e.what(), e.status);
    return 3;
  }
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
    else if (k == "--json") o.json = 1;
    else if (k == "--id-file" && a + 1 < argc) o.id_file = argv[++a];
    else {
      fprintf(stderr, "unknown option %s\n", k.c_str());
      return 2;
    }
  }
  if (o.world < 1 || o.n < o.world || o.topk < 1 || o.steps < 1) return 2;
  if (o.rank >= 0) return run_rank(o);

  // ---- launcher: fork one process per rank; every rank sees ALL GPUs and takes device `rank` (RCCL needs to see its peers for
  // the peer-to-peer transport; narrowing HIP_VISIBLE_DEVICES to one device per rank would push it to host staging)
  int32_t n_dev = 0;
  if (hhv_device_count(&n_dev) != HHV_OK || n_dev < 1) {
    fprintf(stderr, "no HIP device: %s\n", hhv_last_error());
    return 3;
  }
  if (o.world > n_dev) {
    fprintf(stderr, "--world %d needs %d GPUs, %d visible (RCCL does not put two ranks on one device)\n", o.world, o.world, n_dev);
    return 7;
  }
  // the id file lives in a directory of this run's own (mkdtemp: mode 0700, unpredictable name), not at a guessable path in /tmp
  char dir[] = "/tmp/hhv_rccl_XXXXXX";
  if (!mkdtemp(dir)) {
    perror("mkdtemp");
    return 4;
  }
  char idf[256];
  snprintf(idf, sizeof(idf), "%s/id", dir);
  std::vector<pid_t> kids;
  for (int r = 0; r < o.world; ++r) {
    const pid_t pid = fork();
    if (pid == 0) {
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
  unlink((std::string(idf) + ".tmp").c_str());
  rmdir(dir);
  printf("%s\n", rc == 0 ? "sharded_search_rccl: OK" : "sharded_search_rccl: FAILED");
  return rc;
}
