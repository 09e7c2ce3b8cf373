#!/usr/bin/env python3
"""bench.py -- Viterbi DP-cells/s of the MI355X engine on BASELINE.json's workload.

One "step" = one search of the resident database: the query goes to the device (hhv_set_query: 32 KB H2D, inside the timed
step as SURVEY.md 8d asks), the hot path runs (Viterbi::Align for every template of the resident set: hhv_align_async
through the C ABI) over synthetic prepared profiles that already live in HBM, the step's K best records are selected on the
device (hhv_topk) and merged (hhv_merge_hits).  With N > 1 ranks the template database is sharded - ONE global database,
every rank derives the same global length vector, calls hhv_shard_plan(n, L, N) and materialises only its own shard - and
the K best records of every rank are exchanged with ONE all_gather over RCCL (torch.distributed backend "nccl") between
the selection and the merge; torch's stream and the library's stream are ordered with events, the host does not wait inside
a step.

Defaults = BASELINE.json's configs: --gpus 1: Lq 300 vs 100 000 x Lt 300 (the north-star's 1-GPU headline); --gpus 8:
configs[3], 1 000 000 templates = 125 000 per GPU (N = 2, 4: the same 125 000 per GPU); --lengths zipf --gpus N:
configs[4], ONE global Zipf(1.2) length vector 50..1000 cut by hhv_shard_plan (length-binned LPT), the line reports the
stream records of every shard and max / mean.

Prints ONE JSON line on rank 0 (contract in the task description), carrying `roofline` (dominant kernel =
hhv_stream_kernel, timed live with HIP events on the library's stream) and `cpu_baseline` (the reference's own
Viterbi::Align batch loop, oracle/_ref, on the box's usable host cores, bounded sample, thread-count table).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (re-exec under
torch.distributed.run on 127.0.0.1); with fewer GPUs than ranks the ranks share devices and the exchange runs over gloo
(RCCL refuses two ranks on one device) - a correctness path for the tests, flagged `oversubscribed` in the line.
`--force-dist` sends a ONE-rank run through init_process_group("nccl") + all_gather_into_tensor as well (the RCCL branch
on a single GPU: tests/test_gpu_configs.py).

At N = 1 the default run adds the other single-GPU BASELINE configs beside the headline (not part of `value`):
configs1_10k_templates, configs2_backtrace_top500 (10 k and the resident set: backtrace + Hit scores + top-500, checked
against the reference on a sample), configs4_zipf (mixed lengths 50-1000, local mode, checked against the oracle),
template_upload (what the timed region excludes: pack + H2D through hhv_upload_templates, hhv_db_open of the packed file)
and next_rows.
"""
import os

# the CPUs this process may run on, read before an OpenMP runtime exists: with OMP_PROC_BIND set, libgomp binds the initial
# thread to its first place when it is loaded (torch brings one), and sched_getaffinity would report that one core
try:
    AFFINITY_SET = os.sched_getaffinity(0)
    AFFINITY_CPUS = len(AFFINITY_SET)
except AttributeError:
    AFFINITY_SET = None
    AFFINITY_CPUS = os.cpu_count() or 1
# the cpu_baseline leg times OpenMP code (oracle/_ref): pin its threads to cores, one per core, before any OpenMP runtime
# is loaded - unpinned threads of a dynamic schedule migrate and the number is not reproducible
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # streams of the MAC length classes (hhv_create): before HIP initialises
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")

import argparse
import json
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_LANEOPS = 256 * 4 * 32 * 2.4e9  # 256 CU x 4 SIMD-32 x 2.4 GHz = 78.6 T lane-ops/s: one wave64 VALU instruction
                                          # per SIMD every 2 clk.  The guide's 157.3 TFLOP/s vector peak counts an FMA as
                                          # 2 flops on the same issue rate; the reference arithmetic has no FMA
                                          # (separate mul and add are part of the parity contract), so 78.6 T is the bound.
VALU_PEAK_FMA_TFLOPS = 157.3
# measured: two waves per SIMD of a 64-thread / 248-VGPR kernel retire one v_add_f32 wave-instruction per 2.25 clk (nominal
# 2.4 GHz) per SIMD - tools/gen_shape_ubench.py, profiles/r2_shape_ubench.txt
MEASURED_ISSUE_CEILING = 256 * 4 * 64 / 2.25 * 2.4e9
PROFILE_JSONS = ("r6_summary.json", "r5_summary.json", "r4_summary.json", "r3_summary.json", "r2_summary.json")   # newest committed PMC profile of the headline command first
REC_BYTES = 112
OPS_PER_CELL = 92          # SURVEY.md 8d / BASELINE.md 5: fp32 operations per DP cell of the reference
ZIPF_SEED = 0x21F          # the ONE seed of configs[4]'s global length vector


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--templates", type=int, default=None,
                    help="templates per GPU; default 100000 at --gpus 1 (north-star 1-GPU headline), 125000 at --gpus > 1 "
                         "(BASELINE configs[3]: 1 M templates on 8 GPUs)")
    ap.add_argument("--lq", type=int, default=300)
    ap.add_argument("--lt", type=int, default=300)
    ap.add_argument("--lengths", default="fixed", choices=["fixed", "zipf"],
                    help="zipf = BASELINE configs[4]: L_t = 49 + k, k ~ Zipf(1.2) truncated to 50..1000, one global vector")
    ap.add_argument("--topk", type=int, default=500)
    ap.add_argument("--local", type=int, default=0)
    ap.add_argument("--backtrace", type=int, default=0, help="1 = BASELINE configs[2] (backtrace + hit list)")
    ap.add_argument("--ss", type=int, default=0, choices=[0, 1, 2, 4],
                    help="secondary-structure scoring (SURVEY 8a A5, the ...AndSS kernels): 4 PRED_PRED, 2 DSSP_PRED, 1 PRED_DSSP - codes added to "
                         "the stream's meta words, random score tables, the query's codes set per step (tools/bench_rows.py)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the short measurements of the SURVEY 8f rows (N2-N4)")
    ap.add_argument("--no-configs2", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the `pipeline` entry (one hhblits-style search iteration, every stage on the device)")
    ap.add_argument("--pipeline-db", type=int, default=200000, help="sequences of the pipeline entry's resident database (1 000 000: profiles/r6_pipeline.json)")
    ap.add_argument("--no-configs4", action="store_true")
    ap.add_argument("--no-upload", action="store_true", help="skip the template_upload entry (pack + H2D, packed-file open)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the fast_mode entry (the opt-in fused-emission build)")
    ap.add_argument("--no-rows", action="store_true", help="skip ss_modes / multi_strip / masked_round (tools/bench_rows.py: SURVEY 8a rows A5, A4 and "
                    "the multi-strip queries the headline configuration does not exercise)")
    ap.add_argument("--virtual-shards", type=int, default=1,
                    help="single rank only: hold ALL shards of the V-shard database (V x --templates, same global plan, same "
                         "global ids) - the single-process reference of a V-rank run")
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank: still go through torch.distributed (nccl = RCCL) for the hit-list exchange")
    ap.add_argument("--dump-topk", default=None, help="rank 0 writes the merged top-K of the last step (global id, score bits) as .npy")
    if len(sys.argv) == 1 and os.environ.get("HHV_BENCH_ARGV"):   # a rank started by respawn() below
        return ap.parse_args(json.loads(os.environ["HHV_BENCH_ARGV"]))
    return ap.parse_args()


def free_port():
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def respawn(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves, exactly as the driver's launcher does."""
    from pyhhv import capi
    ndev = capi.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus:
        env.setdefault("HHV_BENCH_BACKEND", "gloo")
    # the script's own options travel in the environment: torch.distributed.run's parser would try to match some of them
    # (--local ...) against its own abbreviated options
    env["HHV_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)]
    os.execvpe(cmd[0], cmd, env)


def global_plan(args, capi, synth, n_per, n_shards):
    """The ONE database of a run, identical on every rank: global length vector, hhv_shard_plan over n_shards, and per
    shard the global ids it holds - longest first, the order the reference gives a block before it cuts it into SIMD
    batches (src/hhviterbirunner.cpp:117-119; the batch loop of :122 is the unit of independence the shards split)."""
    n_total = n_per * n_shards
    if args.lengths == "zipf":
        Lg = synth.zipf_lengths(ZIPF_SEED, n_total).astype(np.int32)
    else:
        Lg = np.full(n_total, args.lt, dtype=np.int32)
    shard_of = capi.shard_plan(Lg, n_shards)
    ids = []
    for r in range(n_shards):
        g = np.nonzero(shard_of == r)[0]
        ids.append(g[np.argsort(-Lg[g], kind="stable")].astype(np.int64))
    records = np.array([int(np.sum(Lg[g].astype(np.int64) + 1)) for g in ids], dtype=np.int64)
    return Lg, ids, records


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut by the cgroup CPU quota if there is one
    (os.cpu_count() ignores both)."""
    aff = AFFINITY_CPUS
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    usable = aff if quota is None else max(1, min(aff, int(quota)))
    return usable, aff, quota


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args)   # does not return
    # stdout carries the ONE JSON line and nothing else: libraries that print there (RCCL's version banner at init) are
    # sent to stderr for the life of the process; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    from pyhhv import capi, shard, synth, synth_stream

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    backend = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:   # --force-dist outside torchrun
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        # backend "nccl" IS RCCL on ROCm; gloo only when ranks have to share a device (RCCL refuses two ranks on one
        # device): a correctness path for the tests, never a measurement
        backend = os.environ.get("HHV_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", world_size=world, rank=rank, device_id=device)
        else:
            dist.init_process_group(backend=backend, world_size=world, rank=rank)
    if world > 1 and args.virtual_shards != 1:
        raise SystemExit("--virtual-shards is the single-rank stand-in of a multi-rank run")

    Lq, Lt = args.lq, args.lt
    n = args.templates if args.templates else (100000 if args.gpus == 1 else 125000)   # per GPU
    n_shards = world if args.virtual_shards == 1 else args.virtual_shards
    Lglobal, shard_ids, shard_records = global_plan(args, capi, synth, n, n_shards)
    mine = [rank] if args.virtual_shards == 1 else list(range(n_shards))
    gids = np.concatenate([shard_ids[r] for r in mine])

    qf, qtr = synth_stream.query_np(Lq, synth.PB)   # the stream seeded with 0x51000000
    rec, rec_off, Ls = synth_stream.gen_stream(torch, device, gids, Lglobal[gids], synth.PB)
    n_local = int(Ls.shape[0])
    torch.cuda.synchronize()

    ctx = capi.Context(local=args.local, device=dev_index)
    ctx.set_query(qf, qtr)
    q_ss = None
    if args.ss:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_rows
        bench_rows.add_ss_codes(torch, rec, int(rec_off[-1]))
        ctx.set_ss_tables(*bench_rows.ss_tables())
        q_ss = bench_rows.query_ss(Lq)
        ctx.set_query_ss(*q_ss)
        ctx.set_ss_mode(args.ss)
    ts = ctx.adopt_device_stream(Ls, rec.data_ptr())
    cells_per_rank = ts.cells()
    K = args.topk
    topk_buf = torch.zeros((K, shard.REC_I32), dtype=torch.int32, device=device)   # K hhv_hit records (40 B each)
    ctx.set_global_ids(ts, gids)   # hhv_topk reports global template ids
    gloo = use_dist and dist.get_backend() == "gloo"   # debug path only (HHV_BENCH_BACKEND, ranks sharing a device)
    gathered = torch.zeros((world * K, shard.REC_I32), dtype=torch.int32, device="cpu" if gloo else device)
    merged_buf = torch.zeros((K, shard.REC_I32), dtype=torch.int32, device=device)
    bt = bool(args.backtrace)
    # the library works on its own (non-blocking) stream; torch's collectives on torch's current stream: ordered by events
    lib_stream = torch.cuda.ExternalStream(ctx.stream(), device=device)
    ev_topk = torch.cuda.Event()
    ev_gathered = torch.cuda.Event()

    kernel_ms = []
    ag_events = []   # (start, end) of every timed step's all-gather on torch's stream (read after the timed region)

    def step(record_ms=True):
        ctx.set_query(qf, qtr)   # H2D of the query is part of a search
        if q_ss is not None:
            ctx.set_query_ss(*q_ss)   # (... and so are its secondary-structure codes: hhv_set_query forgets the previous query's)
        ctx.align_async(ts, backtrace=bt)
        if bt:
            ctx.hits(ts, fetch=False)
        ctx.topk(ts, K, d_out=topk_buf.data_ptr(), fetch=False, raw=not bt)
        # hit-list exchange: ONE all_gather of K records per rank over RCCL, then the same device merge on every rank
        # (hhv_merge_hits; with one rank it merges the rank's own list, so that a step is the same job at every N)
        src = topk_buf
        if use_dist and gloo:
            ctx.sync()
            dist.all_gather_into_tensor(gathered, topk_buf.cpu())
            src = gathered.to(device)
            torch.cuda.current_stream().synchronize()
        elif use_dist:
            cur = torch.cuda.current_stream()
            ev_topk.record(lib_stream)
            cur.wait_event(ev_topk)           # the collective starts when this rank's K records are written
            if record_ms:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
            dist.all_gather_into_tensor(gathered, topk_buf)
            if record_ms:
                e1.record(cur)
                ag_events.append((e0, e1))
            ev_gathered.record(cur)
            lib_stream.wait_event(ev_gathered)   # the merge starts when the gathered records have arrived
            src = gathered
        ctx.merge_hits(src.data_ptr(), world * K, K, d_out=merged_buf.data_ptr(), fetch=False, count=False)
        if record_ms:
            kernel_ms.append(ctx.last_kernel_ms())   # (waits for the DP kernel's end event only; the merge is still queued)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    merged = merged_buf[merged_buf[:, shard.COL_INDEX] >= 0]
    # what every rank saw, for an attributable scaling curve: its DP kernel, its all-gather (from the moment its own K records
    # were ready to the arrival of everybody's: includes the wait for the slowest rank) and its wall time of the timed region
    ag_ms = [a.elapsed_time(b) for a, b in ag_events]
    mine = [float(np.mean(kernel_ms)), float(np.min(kernel_ms)), float(np.mean(ag_ms)) if ag_ms else 0.0,
            float(np.min(ag_ms)) if ag_ms else 0.0, dt / args.steps * 1e3, float(n_local), float(cells_per_rank)]
    per_rank = [mine]
    if world > 1:
        allr = torch.zeros((world, len(mine)), dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_gather_into_tensor(allr, torch.tensor([mine], dtype=torch.float64, device=allr.device))
        per_rank = allr.cpu().tolist()
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tcells = torch.tensor([cells_per_rank], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tcells, op=dist.ReduceOp.SUM)
        total_cells = float(tcells.item())
    else:
        total_cells = float(cells_per_rank)
    if args.dump_topk and rank == 0:
        np.save(args.dump_topk, merged.cpu().numpy()[:, [shard.COL_INDEX, 0, 6, 7]])   # global id, score bits, i2, j2

    value = total_cells * args.steps / dt
    k_ms = float(np.mean(kernel_ms))
    algo_bytes = (int(rec_off[-1]) + 1) * REC_BYTES + n_local * 16 + 64 * ((Lq + 63) // 64) * REC_BYTES
    if bt:
        algo_bytes += cells_per_rank  # 1 backtrace byte per cell
    achieved_gbs = algo_bytes / (k_ms * 1e-3) / 1e9
    # SURVEY.md 8(d)'s own figure: 27 floats per column incl. column 0 read, 12 bytes (score, i2, j2) written per template, the
    # query once; + 1 byte per cell written with backtrace.  (The engine's record has 28 dwords - the 28th is the meta word - and
    # its result 16 bytes: `algorithmic_bytes_per_launch` above; the two differ by 3.7 %.)
    algo_bytes_8d = int(rec_off[-1]) * 108 + n_local * 12 + 108 * (Lq + 1) + (cells_per_rank if bt else 0)
    kernel_cells_s = cells_per_rank / (k_ms * 1e-3)

    # HBM traffic and VALU instruction count of the dominant kernel from the committed rocprofv3 PMC profile
    # (profiles/, collected with tools/profile.sh on this same command: separate --pmc passes, FETCH_SIZE doubled as
    # the guide prescribes)
    traffic = None
    valu_wave_instr = None
    profile_json = None
    headline = n_local == 100000 and Lq == 300 and Lt == 300 and not bt and args.lengths == "fixed" and not args.local and not args.ss
    if headline:
        for name in PROFILE_JSONS:
            try:
                with open(os.path.join(ROOT, "profiles", name)) as f:
                    prof = json.load(f)
                traffic = prof.get("traffic_bytes_per_launch")
                valu_wave_instr = prof.get("valu_wave_instr_per_launch")
                profile_json = name
                break
            except Exception:
                continue

    n_global = int(Lglobal.shape[0])
    out = {
        "metric": "viterbi_dp_cells_per_s",
        "value": value,
        "unit": "cells/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "generator_version": 2,   # 2 = SURVEY 8(d) to the letter since round 4 (Gamma(0.5) columns, the prescribed query stream); rounds 1-3 = 1
        "templates_per_s": n_global * args.steps / dt,
        "config": {
            "workload": "Lq%d_vs_%dx_Lt%s_%s_%s" % (Lq, n_global, Lt if args.lengths == "fixed" else "zipf50-1000",
                                                     "local" if args.local else "global",
                                                     ("backtrace_hits_top%d" % K if bt else "score_only_top%d" % K) + ("_ss%d" % args.ss if args.ss else "")),
            "templates_total": n_global, "templates_per_gpu": n if args.virtual_shards == 1 else n_local,
            "Lq": Lq, "Lt": Lt if args.lengths == "fixed" else "zipf(1.2) 50..1000, seed 0x%X" % ZIPF_SEED, "topk": K,
            "parallelism": "template-db-shard x%d (hhv_shard_plan), one %s all_gather of top-K" % (world, "RCCL" if backend == "nccl" else backend)
                           if use_dist else "single GPU",
            "step": "hhv_set_query (H2D) + hhv_align_async + hhv_topk%s + hhv_merge_hits; templates resident in HBM"
                    % (" + all_gather" if use_dist else ""),
            "prng": "splitmix64 -> xoshiro256**, seed 0x5EED0000 + global template id (query 0x51000000), u = (x >> 40) * 2^-24",
            "columns": "SURVEY.md 8(d): f = 0.7 g + 0.3 pb, g = normalised Gamma(0.5) draws (erfinv(2u - 1)^2), p = f / pb; M2I, M2D = "
                       "0.6 log2 U[0.01, 0.05], M2M = log2(1 - pI - pD), I2M = D2M = log2 0.6, I2I = D2D = 0.6 log2 0.4 (pyhhv/synth_stream.py)",
        },
        "roofline": {
            # achieved / frac: SURVEY.md 8(d)'s bytes (27 floats a column, 12 bytes a result); the engine's own records (28 dwords,
            # 16-byte results: 3.7 % more) under achieved_engine_bytes / frac_engine_bytes
            "bound": "hbm", "achieved": algo_bytes_8d / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": algo_bytes_8d / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
            "achieved_engine_bytes": achieved_gbs, "frac_engine_bytes": achieved_gbs / HBM_PEAK_GBS,
            "traffic_source": "profiles/%s (rocprofv3 PMC, bytes per launch)" % profile_json if traffic else None,
            "kernel": "hhv_stream_kernel", "kernel_ms": k_ms, "kernel_ms_min": float(np.min(kernel_ms)),
            "kernel_ms_median": float(np.median(kernel_ms)), "algorithmic_bytes_per_launch": algo_bytes,
            "algorithmic_bytes_8d": algo_bytes_8d, "achieved_8d": algo_bytes_8d / (k_ms * 1e-3) / 1e9,
            "frac_8d": algo_bytes_8d / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "note": "the path is VALU-issue bound, not HBM bound (SURVEY.md 8d): see roofline_valu",
        },
        "roofline_valu": {
            "bound": "valu_fp32_issue", "achieved": kernel_cells_s * OPS_PER_CELL / 1e12, "peak": VALU_PEAK_LANEOPS / 1e12,
            "unit": "T lane-ops/s", "frac": kernel_cells_s * OPS_PER_CELL / VALU_PEAK_LANEOPS,
            "ops_per_cell": OPS_PER_CELL, "kernel_cells_per_s": kernel_cells_s,
            "peak_note": "78.6 T = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz (one non-FMA op per lane per clock); the guide's "
                         "%.1f TFLOP/s vector peak counts FMA as 2 flops on the same issue rate" % VALU_PEAK_FMA_TFLOPS,
            "frac_of_fma_peak": kernel_cells_s * OPS_PER_CELL / (VALU_PEAK_FMA_TFLOPS * 1e12),
        },
    }
    if use_dist:
        out["per_rank"] = [{"rank": r, "dp_kernel_ms": v[0], "dp_kernel_ms_min": v[1], "all_gather_ms": v[2], "all_gather_ms_min": v[3],
                            "ms_per_step": v[4], "templates": int(v[5]), "cells": int(v[6])} for r, v in enumerate(per_rank)]
        out["per_rank_note"] = ("all_gather_ms runs from the moment the rank's own K records are ready to the arrival of every rank's: "
                                "it contains the wait for the slowest rank; dp_kernel_ms is hhv_stream_kernel alone (HIP events on the library's stream)")
    if n_shards > 1:
        out["config"]["shards"] = {"stream_records_per_shard": [int(x) for x in shard_records],
                                   "templates_per_shard": [int(len(g)) for g in shard_ids],
                                   "imbalance_max_over_mean": float(shard_records.max() / shard_records.mean())}
    if world > 1 and ndev < world:
        out["config"]["oversubscribed"] = "%d ranks on %d device(s): correctness path, not a measurement" % (world, ndev)
    if profile_json:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import srchash
            with open(os.path.join(ROOT, "profiles", profile_json)) as f:
                stamp = json.load(f).get("kernel_sources_sha1")
            now = srchash.kernel_sources_sha1(srchash.VITERBI)
            out["roofline"]["profile_sources_sha1"] = stamp
            out["roofline"]["profile_stale"] = (stamp != now) if stamp else "unstamped (taken before round 6)"
        except Exception as e:  # noqa: BLE001
            out["roofline"]["profile_stale"] = "unknown: %r" % (e,)
    if valu_wave_instr:
        # executed VALU instructions from the SQ counters: issue slots used / issue slots available in the kernel's time
        lane_ops = valu_wave_instr * 64.0
        out["roofline_valu"]["counters"] = {
            "valu_wave_instr_per_launch": valu_wave_instr,
            "valu_lane_instr_per_cell": lane_ops / cells_per_rank,
            "achieved_lane_ops_per_s": lane_ops / (k_ms * 1e-3),
            "frac_of_issue_peak": lane_ops / (k_ms * 1e-3) / VALU_PEAK_LANEOPS,
            "source": "profiles/%s (SQ_INSTS_VALU per launch) / kernel_ms of this run" % profile_json,
            # what a SIMD of this chip actually issues: a plain v_add_f32 stream in a kernel of this shape (64-thread
            # workgroups, 248 VGPRs, two waves per SIMD) retires one wave-instruction per 2.23-2.29 clk of the nominal 2.4 GHz
            # (a lone wave one per 4.5), profiles/r2_shape_ubench.txt - the ceiling the kernel's 2 waves per SIMD can reach
            "measured_issue_ceiling_lane_ops_per_s": MEASURED_ISSUE_CEILING,
            "frac_of_measured_issue_ceiling": lane_ops / (k_ms * 1e-3) / MEASURED_ISSUE_CEILING,
        }

    single = rank == 0 and world == 1 and args.virtual_shards == 1
    plain = not bt and args.lengths == "fixed" and not args.ss
    if single and not args.no_configs1 and n >= 10000 and plain:
        # BASELINE configs[1]: the same query vs the first 10k templates of the resident stream
        ts10 = ctx.adopt_device_stream(np.full(10000, Lt, dtype=np.int32), rec.data_ptr())
        ctx.set_global_ids(ts10, np.arange(10000, dtype=np.int32))
        buf10 = torch.zeros((2 * K, shard.REC_I32), dtype=torch.int32, device=device)

        def step10():   # the headline's step on the smaller set: query H2D, DP, top-K, merge of the rank's own list
            ctx.set_query(qf, qtr)
            ctx.align_async(ts10)
            ctx.topk(ts10, K, d_out=buf10.data_ptr(), fetch=False, raw=True)
            ctx.merge_hits(buf10.data_ptr(), K, K, d_out=buf10[K:].data_ptr(), fetch=False, count=False)

        for _ in range(5):
            step10()
        ctx.sync()
        t1 = time.perf_counter()
        reps = 100                  # (a launch that follows an idle device runs ~0.2 ms longer, tools/SESSIONS.md call 43:
        k10 = []                    #  a hundred searches back to back, nothing waited for in between, like the headline loop)
        for _ in range(reps):
            step10()
            k10.append(ctx.last_kernel_ms())   # (waits for the DP kernel's end event only: the top-K and the merge are still queued)
        ctx.sync()
        d10 = time.perf_counter() - t1
        out["configs1_10k_templates"] = {"cells_per_s": 10000 * Lq * Lt * reps / d10, "ms_per_step": d10 / reps * 1e3, "kernel_ms": float(np.mean(k10)),
                                         "steps": reps, "step": "as the headline's: hhv_set_query + hhv_align_async + hhv_topk + hhv_merge_hits"}
        ts10.free()

    if single and not args.no_cpu_baseline:   # the contract: rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args, rec, rec_off, Ls, ctx, ts, qf, qtr, n, Lq)
    elif world > 1:
        out["cpu_baseline"] = None
        out["cpu_baseline_note"] = "timed at N = 1 only (rank 0 of a one-GPU run): see the --gpus 1 line"

    if single and not args.no_configs2 and n >= 10000 and plain:
        out["configs2_backtrace_top500"] = configs2(args, torch, ctx, ts, rec, rec_off, Ls, qf, qtr, Lq, Lt, K)
    if isinstance(out.get("cpu_baseline"), dict):
        # like for like: the reference's Viterbi::Align always writes its backtrace bytes, the GPU headline step does not
        cb = out["cpu_baseline"]
        cb["note"] = ("the CPU leg is Viterbi::Align WITH its backtrace matrix (the reference has no score-only mode); the GPU `value` is the "
                      "score-only step.  The like-for-like ratio is gpu_backtrace_over_cpu (configs2_backtrace_top500, 100 k templates: DP with "
                      "backtrace bytes + walk + Hit scores + top-K)")
        try:
            g = out["configs2_backtrace_top500"]["100k"]["cells_per_s"]
            cb["gpu_backtrace_cells_per_s"] = g
            cb["gpu_backtrace_over_cpu"] = g / cb["value"]
            cb["gpu_score_only_over_cpu"] = value / cb["value"]
        except Exception:
            pass

    if single and not args.no_configs4 and plain:
        out["configs4_zipf"] = configs4(args, torch, capi, synth, device, dev_index, qf, qtr, Lq, K)

    if single and not args.no_upload and plain:
        out["template_upload"] = template_upload(capi, ctx, rec, rec_off, Ls, qf, qtr, Lt)

    if single and not args.no_fast_mode and plain:
        out["fast_mode"] = fast_mode(args, capi, ctx, ts, rec, Ls, qf, qtr, dev_index, K)

    if single and not args.no_rows and plain and n >= 50000:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_rows
        out["ss_modes"] = bench_rows.ss_modes(torch, capi, dev_index, rec, rec_off, Ls, qf, qtr, K, local=args.local)
        out["multi_strip"] = bench_rows.multi_strip(torch, capi, dev_index, rec, rec_off, Ls, K, local=args.local)
        out["masked_round"] = bench_rows.masked_round(torch, capi, dev_index, rec, rec_off, Ls, qf, qtr, K, local=args.local)

    if single and not args.no_next_rows and plain:
        out["next_rows"] = next_rows()
    if single and not args.no_pipeline and plain:
        out["pipeline"] = pipeline(args)

    if single and os.environ.get("HHV_DEBUG_CLK"):
        # measurement builds (-DHHV_EXP_TIMING) export the shader-clock totals of wave 0
        import ctypes
        lib = capi.load()
        if hasattr(lib, "hhv_debug_clk"):
            ctx.align(ts)
            buf = (ctypes.c_ulonglong * 8)()
            lib.hhv_debug_clk(buf)
            out["debug_clk"] = [int(x) for x in buf]
    if use_dist:
        dist.barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ts.free()
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


def fast_mode(args, capi, ctx, ts, rec, Ls, qf, qtr, dev_index, K):
    """OPT-IN, never `value`: the same resident set through libhhviterbi_hip_fma.so (make lib_fma; emission score with fused
    multiply-adds, viterbi_lane.h HHV_EMISSION_FMA) - what it gains and what it changes against the bit-exact default build
    on ALL templates of the set: end points, scores, top-K, and with backtrace the alignments and Hit scores."""
    out = {"library": "libhhviterbi_hip_fma.so", "what": "emission with v_fmac_f32 (16 of the 20 products accumulate fused, log2f4's "
           "polynomial fused): one rounding per term instead of two; opt-in build, the default library has no FMA.  NOT within the 1e-4 "
           "tolerance on every template: log2f4 jumps by 3.93e-4 at powers of two (1.2e7 templates: max |dscore| 3.98e-4, 12 end points "
           "changed - profiles/r6_fast_mode_bound.json)"}
    try:
        cf = capi.Context(local=args.local, device=dev_index, lib_path=capi.FMA_LIB_PATH)
        cf.set_query(qf, qtr)
        tf = cf.adopt_device_stream(Ls, rec.data_ptr())
        for _ in range(2):
            cf.align_async(tf)
            cf.topk(tf, K, fetch=False, raw=True)
        cf.sync()
        reps, ms = 10, []
        t1 = time.perf_counter()
        for _ in range(reps):
            cf.set_query(qf, qtr)
            cf.align_async(tf)
            cf.topk(tf, K, fetch=False, raw=True)
            ms.append(cf.last_kernel_ms())
        cf.sync()
        sec = (time.perf_counter() - t1) / reps
        out["score_only"] = {"cells_per_s": tf.cells() / sec, "ms_per_step": sec * 1e3, "dp_kernel_ms": float(np.mean(ms))}
        f, d = cf.align(tf), ctx.align(ts)
        top_f, _ = cf.topk(tf, K, raw=True)
        top_d, _ = ctx.topk(ts, K, raw=True)
        out["vs_default_build"] = {
            "templates": int(ts.n), "endpoint_mismatches": int(np.sum((f["i2"] != d["i2"]) | (f["j2"] != d["j2"]))),
            "scores_changed": int(np.sum(f["score"] != d["score"])),
            "max_abs_score_diff": float(np.max(np.abs(f["score"].astype(np.float64) - d["score"].astype(np.float64)))),
            "topk_same_templates_same_order": bool(np.array_equal(top_f["index"], top_d["index"]))}
        secb, kmsb = time_bt_steps(cf, tf, K, reps=3, warm=1)
        out["backtrace_hits"] = {"cells_per_s": tf.cells() / secb, "ms_per_step": secb * 1e3, "dp_kernel_ms": kmsb}
        hf = cf.hits(tf)
        ctx.align_async(ts, backtrace=True)
        hd = ctx.hits(ts)
        out["vs_default_build"].update({
            "alignment_mismatches_i1_j1_nsteps_matched_cols": int(np.sum((hf["i1"] != hd["i1"]) | (hf["j1"] != hd["j1"]) |
                                                                         (hf["nsteps"] != hd["nsteps"]) | (hf["matched_cols"] != hd["matched_cols"]))),
            "max_abs_hit_score_diff": float(np.max(np.abs(hf["score"].astype(np.float64) - hd["score"].astype(np.float64))))})
        tf.free()
        cf.close()
    except Exception as e:  # the headline line must not depend on the side measurements
        out["error"] = repr(e)
    return out


def template_upload(capi, ctx, rec, rec_off, Ls, qf, qtr, Lt):
    """What the timed region leaves out ("template upload excluded and reported separately"): the equal-length
    benchmark set handed over as HOST profiles through hhv_upload_templates (pack on the host + H2D), and the same set
    written once with hhv_db_write and loaded with hhv_db_open (validated read + H2D, no packing)."""
    import tempfile
    out = {}
    # the library's packing threads inherit the CPU mask of the thread that calls it - here the initial thread, which libgomp
    # has bound to one core for the cpu_baseline leg (OMP_PROC_BIND above): give it the process's own mask back for this entry
    bound = None
    if AFFINITY_SET is not None:
        try:
            bound = os.sched_getaffinity(0)
            os.sched_setaffinity(0, AFFINITY_SET)
        except OSError:
            bound = None
    try:
        n = int(Ls.shape[0])
        from pyhhv import synth_stream
        host = rec[: int(rec_off[n])].cpu().numpy()
        P, T = synth_stream.unpack_blocks(host, n, Lt)
        del host
        stream_bytes = (int(rec_off[n]) + 1) * REC_BYTES
        t0 = time.perf_counter()
        tsu = ctx.upload_blocks(P, T)
        ctx.sync()
        sec = time.perf_counter() - t0
        out["hhv_upload_templates"] = {"templates": n, "seconds": sec, "stream_bytes": stream_bytes,
                                       "GB_per_s": stream_bytes / sec / 1e9, "what": "host pack (up to 8 threads) into two pinned 64 MiB slabs, H2D of one slab under the packing of the next"}
        want = ctx.align(tsu)
        tsu.free()
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "bench.hhvpdb")
            t0 = time.perf_counter()
            capi.db_write_blocks(path, P, T)
            wsec = time.perf_counter() - t0
            del P, T
            t0 = time.perf_counter()
            tsd = ctx.db_open(path, Ls)
            ctx.sync()
            sec = time.perf_counter() - t0
            got = ctx.align(tsd)
            tsd.free()
        out["hhv_db_open"] = {"templates": n, "seconds": sec, "GB_per_s": stream_bytes / sec / 1e9, "hhv_db_write_seconds": wsec,
                              "what": "packed file (page cache warm) -> read and validated by up to 8 threads into pinned slabs -> H2D; no parsing, no packing",
                              "results_equal_uploaded_set": bool(np.array_equal(got.view(np.uint8), want.view(np.uint8)))}
    except Exception as e:  # the headline line must not depend on the side measurements
        out["error"] = repr(e)
    if bound is not None:
        try:
            os.sched_setaffinity(0, bound)
        except OSError:
            pass
    return out


def time_bt_steps(ctx, ts, K, reps=8, warm=2):
    """Whole step with backtrace: DP kernel with compare bits, trace + rescoring kernels, device top-K by Hit.score."""
    ms = []
    for _ in range(warm):
        ctx.align_async(ts, backtrace=True)
        ctx.hits(ts, fetch=False)
        ctx.topk(ts, K, fetch=False)
    ctx.sync()
    t1 = time.perf_counter()
    for _ in range(reps):
        ctx.align_async(ts, backtrace=True)
        ctx.hits(ts, fetch=False)
        ctx.topk(ts, K, fetch=False)
        ms.append(ctx.last_kernel_ms())
    ctx.sync()
    return (time.perf_counter() - t1) / reps, float(np.mean(ms))


def check_hits_against_cpu(ctx, ts, eng, par, qf, qtr, tps, ttrs, cores, K, replicate):
    """GPU hits, paths and top-K of the first len(tps) templates of `ts` (already aligned with backtrace + hhv_hits)
    against the reference / the oracle: bit-exact indices, equal Hit scores, equal path checksums, same top-K order."""
    import pyoracle
    m = len(tps)
    ref = eng.bench_hits(par, qf, qtr, tps, ttrs, threads=cores, replicate=replicate)
    hits = ctx.hits(ts)
    off, pi, pj, pst, pS = ctx.hit_path_pool(ts)
    ph, sh = pyoracle.path_hashes(off, pi, pj, pst, pS, hits["nsteps"][:m])
    g = hits[:m]
    ok_idx = all(bool(np.array_equal(g[k], ref[k])) for k in ("i1", "j1", "i2", "j2", "nsteps", "matched_cols"))
    ok_vit = bool(np.array_equal(g["viterbi_score"], ref["score"]))
    ok_hit = bool(np.array_equal(g["score"], ref["hit_score"]))
    ok_path = bool(np.array_equal(ph, ref["path_hash"]) and np.array_equal(sh, ref["s_hash"]))
    want = sorted(range(m), key=lambda k: (-float(ref["hit_score"][k]), k))[:K]
    got = sorted(range(m), key=lambda k: (-float(g["score"][k]), k))[:K]
    return {"templates_checked": m, "endpoints_nsteps_matched_cols_bit_exact": ok_idx, "viterbi_scores_equal": ok_vit,
            "hit_scores_equal": ok_hit, "paths_and_S_checksums_equal": ok_path, "topk_order_equal": want == got,
            "max_abs_hit_score_diff": float(np.max(np.abs(g["score"].astype(np.float64) - ref["hit_score"].astype(np.float64)))),
            "cpu_seconds": ref["sec"]}


def configs2(args, torch, ctx, ts, rec, rec_off, Ls, qf, qtr, Lq, Lt, K):
    """BASELINE configs[2]: backtrace + Hit scores (ScoreForBacktrace) + top-500 by Hit.score, 10 k templates and the whole
    resident set; the 10 k run is compared with the reference's own batch loop (Align + Backtrace + ScoreForBacktrace)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from pyhhv import synth_stream
    out = {}
    n10 = 10000
    ts10 = ctx.adopt_device_stream(np.full(n10, Lt, dtype=np.int32), rec.data_ptr())
    sec, kms = time_bt_steps(ctx, ts10, K, reps=20)
    out["10k"] = {"cells_per_s": n10 * Lq * Lt / sec, "ms_per_step": sec * 1e3, "dp_kernel_ms": kms}
    try:
        cores = usable_cores()[0]
        par = pyoracle.make_params(local=args.local)
        use_ref = pyoracle.have_ref()
        eng = pyoracle.Ref() if use_ref else pyoracle.Oracle()
        m = n10 if use_ref else 512
        host = rec[: int(rec_off[m])].cpu().numpy()
        tps, ttrs = synth_stream.unpack_templates(host, rec_off, Ls, m)
        chk = check_hits_against_cpu(ctx, ts10, eng, par, qf, qtr, tps, ttrs, cores, K, replicate=False)
        chk["cpu_kind"] = "reference" if use_ref else "port"
        out["10k"]["gpu_matches_cpu_on_sample"] = chk
    except Exception as e:  # the headline line must not depend on the side measurements
        out["10k"]["gpu_matches_cpu_on_sample"] = {"error": repr(e)}
    ts10.free()
    try:
        sec, kms = time_bt_steps(ctx, ts, K, reps=3, warm=1)
        e = {"cells_per_s": ts.cells() / sec, "ms_per_step": sec * 1e3, "dp_kernel_ms": kms,
             "backtrace_bytes_written_per_launch": int(ts.records()) * 512}
        if ts.n == 100000 and Lq == 300 and Lt == 300:
            try:   # the backtrace kernel's executed VALU instructions and HBM traffic from its own committed counters
                bt_json = next(n for n in ("r5bt_summary.json", "r4bt_summary.json", "r3bt_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
                with open(os.path.join(ROOT, "profiles", bt_json)) as f:
                    prof = json.load(f)
                lane_ops = prof["valu_wave_instr_per_launch"] * 64.0
                e["roofline_valu"] = {"valu_wave_instr_per_launch": prof["valu_wave_instr_per_launch"],
                                      "valu_lane_instr_per_cell": lane_ops / ts.cells(),
                                      "frac_of_issue_peak": lane_ops / (kms * 1e-3) / VALU_PEAK_LANEOPS,
                                      "frac_reference_flops": ts.cells() / (kms * 1e-3) * OPS_PER_CELL / VALU_PEAK_LANEOPS,
                                      "source": "profiles/%s (SQ_INSTS_VALU per launch) / dp_kernel_ms of this run" % bt_json}
                algo = int(ts.records()) * 108 + ts.n * 12 + int(ts.cells())   # 108 B per column read, 12 B per result, 1 B per cell written
                e["roofline_hbm"] = {"algorithmic_bytes_8d_per_launch": algo,
                                     "achieved_GBs_algorithmic": algo / (kms * 1e-3) / 1e9,
                                     "frac_of_hbm_peak_algorithmic": algo / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "counter_traffic_bytes_per_launch": prof["traffic_bytes_per_launch"],
                                     "counter_traffic_over_algorithmic": prof["traffic_bytes_per_launch"] / algo,
                                     "achieved_GBs_counter_traffic": prof["traffic_bytes_per_launch"] / (kms * 1e-3) / 1e9,
                                     "frac_of_hbm_peak_counter_traffic": prof["traffic_bytes_per_launch"] / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "note": "the engine stores one 8-byte compare-bit entry per 5 cells and 320 rows for 300: 1.26 x the algorithmic bytes by design"}
            except Exception:
                pass
        out["%dk" % (ts.n // 1000)] = e
    except Exception as e:
        out["%dk" % (ts.n // 1000)] = {"error": repr(e)}
    return out


def configs4(args, torch, capi, synth, device, dev_index, qf, qtr, Lq, K):
    """BASELINE configs[4] on one GPU: mixed template lengths L_t = 49 + k, k ~ Zipf(1.2) truncated to 50..1000, local mode
    (global + mixed SIMD batches hits the reference's batch-composition quirk), score-only + top-K;
    a sample is compared with the oracle (single-length batches)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from pyhhv import synth_stream
    nz = 200000
    Lz = synth.zipf_lengths(ZIPF_SEED, nz).astype(np.int32)   # the first 200 k templates of configs[4]'s global database
    recz, offz, Lz = synth_stream.gen_stream(torch, device, np.arange(nz), Lz, synth.PB)
    torch.cuda.synchronize()
    c = capi.Context(local=1, device=dev_index)
    c.set_query(qf, qtr)
    tz = c.adopt_device_stream(Lz, recz.data_ptr())
    for _ in range(2):
        c.align_async(tz)
        c.topk(tz, K, fetch=False, raw=True)
    c.sync()
    reps, ms = 5, []
    t1 = time.perf_counter()
    for _ in range(reps):
        c.align_async(tz)
        c.topk(tz, K, fetch=False, raw=True)
        ms.append(c.last_kernel_ms())
    c.sync()
    sec = (time.perf_counter() - t1) / reps
    out = {"templates": nz, "mean_Lt": float(np.mean(Lz)), "max_Lt": int(np.max(Lz)), "mode": "local",
           "cells_per_s": tz.cells() / sec, "templates_per_s": nz / sec, "ms_per_step": sec * 1e3, "dp_kernel_ms": float(np.mean(ms))}
    try:
        m = 4096
        host = recz[: int(offz[m])].cpu().numpy()
        tps, ttrs = synth_stream.unpack_templates(host, offz, Lz, m)
        par = pyoracle.make_params(local=1)
        r = pyoracle.Oracle().bench_align(par, qf, qtr, tps, ttrs, threads=usable_cores()[0])
        gpu = c.align(tz)
        out["gpu_matches_oracle_on_sample"] = {
            "templates_checked": m,
            "endpoints_bit_exact": bool(np.array_equal(gpu["i2"][:m], r[2]) and np.array_equal(gpu["j2"][:m], r[3])),
            "scores_equal": bool(np.all(gpu["score"][:m] == r[1]))}
    except Exception as e:
        out["gpu_matches_oracle_on_sample"] = {"error": repr(e)}
    tz.free()
    c.close()
    del recz
    return out


def next_rows():
    """Short measurements of the callers either side of the path, each checked against the oracle / the
    reference on a sample: prefilter kernels (N3) and MAC realignment (N4).  Not part of `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    out = {}
    try:
        import bench_prefilter
        cores = usable_cores()[0]
        r = bench_prefilter.run(200000, 300, 1000, ref_threads=cores)
        ref = r.get("reference") or {}
        out["N3_prefilter"] = {"db_sequences": r["n_db"], "db_residues": r["residues"], "Lq": r["Lq"],
                               # the reference's AVX2 kernels with its own OpenMP loop over the database, on the host cores of this box
                               "reference_gapless_cells_per_s": ref.get("gapless_cells_per_s"), "reference_sw_cells_per_s": ref.get("sw_cells_per_s"),
                               "reference_cores": ref.get("threads"), "reference_kind": "reference (oracle/_ref, Prefilter::ungapped_sse_score / swStripedByte, AVX2 %s-byte vectors)" % ref.get("vector_bytes"),
                               "reference_error": ref.get("error"),
                               "gapless_cells_per_s": r["ungapped"]["cells_per_s"], "gapless_kernel_ms": r["ungapped"]["kernel_ms"],
                               "sw_cells_per_s": r["gapped"]["cells_per_s"], "sw_kernel_ms": r["gapped"]["kernel_ms"],
                               "mismatches_vs_oracle": r["ungapped"]["mismatches"] + r["gapped"]["mismatches"],
                               "checked": r["ungapped"]["checked"] + r["gapped"]["checked"]}
    except Exception as e:  # the headline line must not depend on the side measurements
        out["N3_prefilter"] = {"error": repr(e)}
    try:
        # Smith-Waterman on what a search sends through it - the survivors of the gapless stage, i.e. homologs - beside random
        # sequences: the lazy-F correction (src/hhprefilter.cpp:176-203) is where the two differ (tools/bench_sw_homologs.py)
        import bench_sw_homologs
        h = bench_sw_homologs.run(20000, usable_cores()[0], 300, 200)
        if isinstance(out.get("N3_prefilter"), dict):
            out["N3_prefilter"]["sw_on_homologs_20000x300"] = {k: h[k] for k in ("random", "homologs_30pct", "homologs_70pct")}
    except Exception as e:
        out.setdefault("N3_prefilter", {})["sw_on_homologs_error"] = repr(e)
    try:
        import bench_mac
        r = bench_mac.run(500, 300, 300, 100, ref_threads=usable_cores()[0])
        ro = r.get("reference_openmp") or {}
        out["N4_mac_realign"] = {"hits": r["n_hits"], "Lq": r["Lq"], "Lt": r["Lt"], "kernels_ms": r["gpu_kernels_ms"],
                                 "end_to_end_ms_host_staged_profiles": r["gpu_wall_ms_incl_host_masks"],
                                 "end_to_end_ms_resident_set": r["resident_set"]["runner_ms"],
                                 "resident_identical_to_staged": r["resident_set"]["identical_to_staged"],
                                 "hits_per_s": r["gpu_hits_per_s"],
                                 "reference_hits_per_s_1core": r.get("ref_cpu_hits_per_s_1core"),
                                 # PosteriorDecoderRunner's OpenMP loop over the templates (src/hhposteriordecoderrunner.cpp:76) on this box's cores
                                 "reference_hits_per_s": ro.get("hits_per_s"), "reference_cores": ro.get("threads"), "reference_ms": ro.get("ms"),
                                 "reference_error": ro.get("error"),
                                 "mismatches_vs_reference": r.get("mismatches_vs_reference"), "checked": r.get("checked")}
    except Exception as e:
        out["N4_mac_realign"] = {"error": repr(e)}
    try:
        import bench_prepare
        out["N2_prepare"] = bench_prepare.run(100000, ref_sample=256, ref_threads=usable_cores()[0])
    except Exception as e:
        out["N2_prepare"] = {"error": repr(e)}
    try:
        # VALU-issue roofline of these kernels from the committed counters (profiles/r5_next_rows_summary.json - r3's if absent -, tools/profile_next.sh):
        # executed VALU lane-instructions per cell x the rate measured here (prefilter), issue fraction of the profiled launch (MAC)
        nr_name = next(n for n in ("r6_next_rows_summary.json", "r5_next_rows_summary.json", "r3_next_rows_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", n)))
        with open(os.path.join(ROOT, "profiles", nr_name)) as f:
            k = json.load(f)["kernels"]
        rv = {}
        for name, key, rate in (("gapless", "hhv_pf_ungapped_kernel", out.get("N3_prefilter", {}).get("gapless_cells_per_s")),
                                ("smith_waterman", "hhv_pf_sw_kernel", out.get("N3_prefilter", {}).get("sw_cells_per_s"))):
            e = next((v for n, v in k.items() if key in n and v.get("valu_lane_instr_per_cell")), None)
            if e and rate:
                rv[name] = {"valu_lane_instr_per_cell": e["valu_lane_instr_per_cell"],
                            "frac_of_valu_issue_peak": rate * e["valu_lane_instr_per_cell"] / VALU_PEAK_LANEOPS,
                            "frac_in_profiled_launch": e["frac_of_valu_issue_peak"]}
        for n, v in k.items():
            if "mac_" in n:
                rv[n.split("<")[0]] = {"frac_in_profiled_launch": v["frac_of_valu_issue_peak"], "avg_ms_profiled": v["avg_ms"]}
        stale = None
        try:
            import srchash
            stamp = json.load(open(os.path.join(ROOT, "profiles", nr_name))).get("kernel_sources_sha1")
            stale = (stamp != srchash.kernel_sources_sha1(srchash.NEXT_ROWS)) if stamp else "unstamped (taken before round 6)"
        except Exception:
            pass
        out["roofline_valu"] = {"bound": "valu_issue", "peak_T_lane_ops_per_s": VALU_PEAK_LANEOPS / 1e12, "kernels": rv, "profile_stale": stale,
                                "source": "profiles/%s (SQ_INSTS_VALU per launch) x the rates of this run" % nr_name}
    except Exception as e:
        out["roofline_valu"] = {"error": repr(e)}
    try:
        # the boundary itself: ViterbiRunner::alignment of the reference (its own translation unit, all host cores it asks
        # for) against the drop-in translation unit, wall time of the call, hits compared (tools/bench_dropin.py)
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libhhref_dropin.so")):
            import bench_dropin
            # 20 000 templates: hhblits' maxnumdb, what one of its rounds hands to ViterbiRunner::alignment (src/hhdecl.cpp:13)
            r = bench_dropin.run(20000, min(32, usable_cores()[0]), 300, altalis=(1,), phases=True)   # (threads the box really grants: 32 on a 16-CPU quota was noise)
            a = r["altali1"]
            out["dropin_ViterbiRunner_alignment"] = {"templates": r["n_templates"], "Lq": r["L"], "Lt": r["L"], "host_threads": r["threads"],
                                                     "reference_s": a["reference_s"], "dropin_cold_cache_s": a["dropin_cold_cache_s"],
                                                     "dropin_warm_cache_s": a["dropin_warm_cache_s"], "dropin_warm_calls_s": a.get("dropin_warm_calls_s"), "hits_identical": a["hits_identical"],
                                                     "cold_phases_ms": a.get("cold_phases_ms"), "warm_phases_ms": a.get("warm_phases_ms"),
                                                     "note": "cold = first call of the process: every template's HHM text parsed by the host's HMM::Read, as in the "
                                                             "reference ('read+parse'), context creation and the first launches' code-object loads included; warm = "
                                                             "templates resident on the device (second search of the process); phases: HHV_DROPIN_TIMING, "
                                                             "hh-suite_amd/dropin/hhviterbirunner_hip.cpp PhaseTimer"}
    except Exception as e:
        out["dropin_ViterbiRunner_alignment"] = {"error": repr(e)}
    try:
        # cold PROCESSES of the reference's hhsearch and of the same program with the replaced translation units: the
        # number the sidecar (N1 inside the product's cold path) exists for is `dropin_cold_with_sidecar_process_s`
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "hhsearch_hip")):
            import bench_apps
            out["dropin_cold_process"] = bench_apps.sidecar_processes(4000, min(16, usable_cores()[0]), 300)
    except Exception as e:
        out["dropin_cold_process"] = {"error": repr(e)}
    return out


def pipeline(args):
    """One hhblits-style search iteration with every stage on the device (what HHblits::run does per round,
    src/hhblits.cpp:1065-1415): gapless prefilter over the resident cs219 database -> Smith-Waterman + second selection ->
    hhv_prepare_subset of the survivors from the resident raw HMMs -> Viterbi + backtrace + Hit scores -> top 500 -> MAC
    realignment.  Per-stage wall times of the best of three iterations (tools/bench_pipeline.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import bench_pipeline
        n_db = int(args.pipeline_db)
        return bench_pipeline.run(n_db, 20000 if n_db >= 1000000 else 10000, 500)
    except Exception as e:  # noqa: BLE001
        return {"error": repr(e)}


def cpu_baseline(args, rec, rec_off, Ls, ctx, ts, qf, qtr, n, Lq):
    """The reference's own batch loop (Viterbi::Align, 8 AVX2 lanes per call, OpenMP dynamic over batches as in
    src/hhviterbirunner.cpp:122) on a bounded sample of the SAME templates, on the host cores this process may use.
    `cores` = the thread count of the reported value: the best of a small table (1, 8, 32, usable), each entry timed on a
    sample sized for a few seconds, threads pinned one per core (OMP_PROC_BIND=close, OMP_PLACES=cores, set at the top of
    this file); the best entry is timed twice.  Also cross-checks the GPU results of the largest sample against it
    (bit-exact endpoints, equal scores)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    from pyhhv import synth_stream
    usable, affinity, quota = usable_cores()
    par = pyoracle.make_params(local=args.local)
    use_ref = pyoracle.have_ref()
    eng = pyoracle.Ref() if use_ref else pyoracle.Oracle()
    Lmean = float(np.mean(Ls))
    if args.lengths != "fixed" and not args.local and use_ref:
        # global mode + mixed-length SIMD batches hits the reference's batch-composition quirk;
        # the parity definition there is the single-length batch = the restatement
        eng, use_ref = pyoracle.Oracle(), False
    cap = min(n, 100000)
    host = rec[: int(rec_off[cap])].cpu().numpy()
    tps, ttrs = synth_stream.unpack_templates(host, rec_off, Ls, cap)
    del host

    def run(threads, m):
        m = max(8, min(cap, m - m % 8))
        r = eng.bench_align(par, qf, qtr, tps[:m], ttrs[:m], threads=threads)
        cells = float(Lq) * float(np.sum(Ls[:m]))
        return {"threads": threads, "templates": m, "seconds": r[0], "cells_per_s": cells / max(r[0], 1e-9)}, r

    # one-thread rate from a short probe sizes every other sample
    probe, _ = run(1, 256)
    rate1 = probe["cells_per_s"]
    per_entry = max(1.0, args.cpu_seconds / 6.0)       # seconds of wall time per table entry
    counts = sorted(set(t for t in (1, 8, 32, usable) if 1 <= t <= usable))
    table = []
    for t in counts:
        m = int(per_entry * rate1 * min(t, 16) / (Lq * Lmean))   # (scaling is far from linear: do not oversize the big runs)
        e, _ = run(t, max(m, 8 * t * 4))
        table.append(e)
    best = max(table, key=lambda e: e["cells_per_s"])
    # the reported value: the best thread count, timed twice on a sample sized from its own measured rate
    m = int(args.cpu_seconds / 3.0 * best["cells_per_s"] / (Lq * Lmean))
    a, ra = run(best["threads"], m)
    b, rb = run(best["threads"], m)
    final = a if a["cells_per_s"] >= b["cells_per_s"] else b
    r = ra
    sample = a["templates"]
    score, i2, j2 = r[-3], r[-2], r[-1]
    gpu = ctx.align(ts)
    ok_idx = bool(np.array_equal(gpu["i2"][:sample], i2) and np.array_equal(gpu["j2"][:sample], j2))
    ok_score = bool(np.all(gpu["score"][:sample] == score))
    maxdiff = float(np.max(np.abs(gpu["score"][:sample].astype(np.float64) - score.astype(np.float64))))
    one = [e for e in table if e["threads"] == 1][0]
    return {
        "value": final["cells_per_s"], "unit": "cells/s", "cores": final["threads"],
        "kind": "reference" if use_ref else "port",
        "sample": "first %d templates of the benchmark set (Lq=%d, mean Lt=%.0f), %s, OpenMP dynamic over batches, "
                  "%d pinned threads, %.2f s (best of two runs; the other: %.3e cells/s)" % (
                      sample, Lq, Lmean, "Viterbi::Align AVX2 8 lanes/call" if use_ref else "scalar C restatement",
                      final["threads"], final["seconds"], min(a["cells_per_s"], b["cells_per_s"])),
        "repeat_ratio": min(a["cells_per_s"], b["cells_per_s"]) / max(a["cells_per_s"], b["cells_per_s"]),
        "host": {"usable_cores": usable, "affinity_mask_cpus": affinity, "cgroup_cpu_quota": quota, "os_cpu_count": os.cpu_count(),
                 "omp_proc_bind": os.environ.get("OMP_PROC_BIND"), "omp_places": os.environ.get("OMP_PLACES")},
        "thread_table": table,
        "scaling_efficiency_vs_1_thread": final["cells_per_s"] / (final["threads"] * one["cells_per_s"]),
        "single_thread_cells_per_s": one["cells_per_s"],
        "gpu_matches_cpu_on_sample": {"endpoints_bit_exact": ok_idx, "scores_equal": ok_score, "max_abs_score_diff": maxdiff},
    }


if __name__ == "__main__":
    main()
