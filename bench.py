#!/usr/bin/env python3
"""bench.py -- Viterbi DP-cells/s of the MI355X engine on BASELINE.json's workload.

One "step" = one pass of the hot path (Viterbi::Align for every template of the resident set:
hhv_align_async through the C ABI) over one batch of synthetic prepared profiles that already live
in HBM, followed by the device-side top-K of the step's results; with N > 1 ranks the template
database is sharded (weak scaling: --templates per GPU) and the K best records of every rank are
exchanged with ONE all_gather over RCCL (torch.distributed backend "nccl").

Prints ONE JSON line on rank 0 (contract in the task description), carrying `roofline` (dominant
kernel = hhv_stream_kernel, timed live with HIP events on the library's stream) and `cpu_baseline`
(the reference's own Viterbi::Align batch loop, oracle/_ref, on the box's host cores, bounded sample).

`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (re-exec under
torch.distributed.run on 127.0.0.1); with fewer GPUs than ranks the ranks share devices and the exchange runs over gloo
(RCCL refuses two ranks on one device) - a correctness path for the tests, flagged `oversubscribed` in the line.

At N = 1 the default run adds the other single-GPU BASELINE configs beside the headline (not part of `value`):
configs1_10k_templates, configs2_backtrace_top500 (10 k and the resident set: backtrace + Hit scores + top-500, checked
against the reference on a sample) and configs4_zipf (mixed lengths 50-1000, local mode, checked against the oracle).
"""
import argparse
import json
import os
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
PROFILE_JSON = "r2_summary.json"
REC_BYTES = 112
OPS_PER_CELL = 92          # SURVEY.md 8d / BASELINE.md 5: fp32 operations per DP cell of the reference


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--templates", type=int, default=100000, help="templates per GPU (north-star 1-GPU headline: 100k)")
    ap.add_argument("--lq", type=int, default=300)
    ap.add_argument("--lt", type=int, default=300)
    ap.add_argument("--lengths", default="fixed", choices=["fixed", "zipf"],
                    help="zipf = BASELINE configs[4]: L_t = 49 + k, k ~ Zipf(1.2) truncated to 50..1000")
    ap.add_argument("--topk", type=int, default=500)
    ap.add_argument("--local", type=int, default=0)
    ap.add_argument("--backtrace", type=int, default=0, help="1 = BASELINE configs[2] (backtrace + hit list)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs1", action="store_true")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the short measurements of the SURVEY 8f rows (N2-N4)")
    ap.add_argument("--no-configs2", action="store_true")
    ap.add_argument("--no-configs4", action="store_true")
    ap.add_argument("--virtual-shards", type=int, default=1,
                    help="single rank only: build the database as the concatenation of the shards V ranks would hold "
                         "(--templates each, same seeds, same global ids) - the single-process reference of a V-rank run")
    ap.add_argument("--dump-topk", default=None, help="rank 0 writes the merged top-K of the last step (global id, score bits) as .npy")
    if len(sys.argv) == 1 and os.environ.get("HHV_BENCH_ARGV"):   # a rank started by respawn() below
        return ap.parse_args(json.loads(os.environ["HHV_BENCH_ARGV"]))
    return ap.parse_args()


def respawn(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves, exactly as the driver's launcher does."""
    import socket
    from pyhhv import capi
    ndev = capi.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ndev < args.gpus:
        env.setdefault("HHV_BENCH_BACKEND", "gloo")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    # the script's own options travel in the environment: torch.distributed.run's parser would try to match some of them
    # (--local ...) against its own abbreviated options
    env["HHV_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    os.execvpe(cmd[0], cmd, env)


def gen_stream(torch, device, Ls_parts, seeds, pb):
    """Synthetic prepared templates generated on the GPU straight into the packed record stream
    (DESIGN.md section 2): per template a header + L[k] column records; + terminal header + pad.
    Ls_parts / seeds: one entry per shard piece (a rank's database is one piece; --virtual-shards concatenates the
    pieces V ranks would hold: the column values of a piece depend only on its own seed and lengths).
    Works for ragged lengths (config 5).  Same distribution family as pyhhv/synth.py (peaky columns mixed
    with the background, divided by the null model; transitions like AddTransitionPseudocounts leaves them).
    Returns (records tensor, rec_off int64 numpy, Ls int32 numpy)."""
    Ls = np.concatenate([np.asarray(x, dtype=np.int64) for x in Ls_parts])
    n = Ls.shape[0]
    rec_off = np.zeros(n + 1, dtype=np.int64)
    rec_off[1:] = np.cumsum(Ls + 1)
    nrec = int(rec_off[-1])
    total = nrec + 1 + 256
    rec = torch.zeros((total, 28), dtype=torch.float32, device=device)
    meta = rec.view(torch.int32)
    pbt = torch.tensor(pb, dtype=torch.float32, device=device)
    off_t = torch.from_numpy(rec_off).to(device)
    L_t = torch.from_numpy(Ls).to(device)
    chunk = 1 << 21
    base = 0
    for Lp, seed in zip(Ls_parts, seeds):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        nrec_p = int(np.sum(np.asarray(Lp, dtype=np.int64) + 1))
        for a0 in range(0, nrec_p, chunk):
            a, b = base + a0, base + min(nrec_p, a0 + chunk)
            m = b - a
            u = torch.rand((m, 20), generator=g, device=device)
            gg = u.pow(6.0) + 1e-9
            gg = gg / gg.sum(dim=1, keepdim=True)
            f = 0.7 * gg + 0.3 * pbt
            f = f / f.sum(dim=1, keepdim=True)
            rec[a:b, 0:20] = f / pbt
            t = torch.rand((m, 8), generator=g, device=device)
            # record j: tr[j-1][M2M,M2D,D2M,D2D,I2M] from the "previous column" draws, tr[j][I2I,M2I] from its own
            pI, pD, pII, pDD = 0.01 + 0.04 * t[:, 0], 0.01 + 0.04 * t[:, 1], 0.25 + 0.3 * t[:, 2], 0.25 + 0.3 * t[:, 3]
            rec[a:b, 20] = torch.log2(1.0 - pI - pD)
            rec[a:b, 21] = torch.log2(pD) * 0.6
            rec[a:b, 22] = torch.log2(1.0 - pDD)
            rec[a:b, 23] = torch.log2(pDD) * 0.6
            rec[a:b, 24] = torch.log2(1.0 - pII)
            rec[a:b, 25] = torch.log2(0.25 + 0.3 * t[:, 4]) * 0.6
            rec[a:b, 26] = torch.log2(0.01 + 0.04 * t[:, 5]) * 0.6
            del u, gg, f, t
        base += nrec_p
    # per-record template id and column index
    pos = torch.arange(nrec, dtype=torch.int64, device=device)
    tid = torch.searchsorted(off_t, pos, right=True) - 1
    j = (pos - off_t[tid]).to(torch.int32)
    Lr = L_t[tid].to(torch.int32)
    is_hdr = j == 0
    # column 1 carries tr[0]: M2M = 0, no M->D out of column 0; column L: no M->I out of L (src/hhhmm.cpp:1755-1785)
    first = j == 1
    rec[:nrec, 20][first] = 0.0
    rec[:nrec, 21][first] = -100000.0
    last = j == Lr
    rec[:nrec, 26][last] = -100000.0
    meta[:nrec, 27] = torch.where(last, j | 0x40000000, j)
    rec[:nrec][is_hdr] = 0.0
    meta[:nrec, 27][is_hdr] = -2 ** 31
    meta[:nrec, 0][is_hdr] = tid[is_hdr].to(torch.int32)
    meta[:nrec, 1][is_hdr] = Lr[is_hdr]
    meta[nrec, 27] = -2 ** 31
    meta[nrec, 0] = -1
    return rec, rec_off, Ls.astype(np.int32)


def unpack_templates(rec_host, rec_off, Ls, n):
    """Packed records (host numpy, first n templates) -> prepared AoS profiles (p[(L+1),20], tr[(L+1),7])
    holding every value the DP reads."""
    tps, ttrs = [], []
    for k in range(n):
        Lt = int(Ls[k])
        body = rec_host[int(rec_off[k]): int(rec_off[k]) + Lt + 1]
        p = np.zeros((Lt + 1, 20), dtype=np.float32)
        tr = np.zeros((Lt + 1, 7), dtype=np.float32)
        p[1:] = body[1:, 0:20]
        tr[:Lt, 0] = body[1:, 20]
        tr[:Lt, 2] = body[1:, 21]
        tr[:Lt, 5] = body[1:, 22]
        tr[:Lt, 6] = body[1:, 23]
        tr[:Lt, 3] = body[1:, 24]
        tr[1:, 4] = body[1:, 25]
        tr[1:, 1] = body[1:, 26]
        tps.append(p)
        ttrs.append(tr)
    return tps, ttrs


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(args)   # does not return
    import torch
    from pyhhv import capi, shard, synth

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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # backend "nccl" IS RCCL on ROCm; gloo only when ranks have to share a device (RCCL refuses two ranks on one
        # device): a correctness path for the tests, never a measurement
        backend = os.environ.get("HHV_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", world_size=world, rank=rank, device_id=device)
        else:
            dist.init_process_group(backend=backend, world_size=world, rank=rank)
    if world > 1 and args.virtual_shards != 1:
        raise SystemExit("--virtual-shards is the single-rank stand-in of a multi-rank run")

    Lq, Lt, n = args.lq, args.lt, args.templates

    def shard_lengths(r):
        if args.lengths == "zipf":
            return synth.zipf_lengths(0x21F + r, n).astype(np.int32)
        return np.full(n, Lt, dtype=np.int32)

    qf, qtr = synth.make_query(0x51000000, Lq)
    pieces = [rank] if args.virtual_shards == 1 else list(range(args.virtual_shards))
    rec, rec_off, Ls = gen_stream(torch, device, [shard_lengths(r) for r in pieces], [0x5EED0000 + r for r in pieces], synth.PB)
    n_local = int(Ls.shape[0])
    torch.cuda.synchronize()

    ctx = capi.Context(local=args.local, device=dev_index)
    ctx.set_query(qf, qtr)
    ts = ctx.adopt_device_stream(Ls, rec.data_ptr())
    cells_per_rank = ts.cells()
    K = args.topk
    topk_buf = torch.zeros((K, shard.REC_I32), dtype=torch.int32, device=device)   # K hhv_hit records (40 B each)
    ctx.set_global_ids(ts, np.arange(n_local, dtype=np.int64) + pieces[0] * n)   # hhv_topk reports global template ids
    gloo = world > 1 and dist.get_backend() == "gloo"   # debug path only (HHV_BENCH_BACKEND, ranks sharing a device)
    gathered = torch.zeros((world * K, shard.REC_I32), dtype=torch.int32, device="cpu" if gloo else device)
    merged_buf = torch.zeros((K, shard.REC_I32), dtype=torch.int32, device=device)
    bt = bool(args.backtrace)

    kernel_ms = []

    def step():
        ctx.align_async(ts, backtrace=bt)
        if bt:
            ctx.hits(ts, fetch=False)
        ctx.topk(ts, K, d_out=topk_buf.data_ptr(), fetch=False, raw=not bt)
        kernel_ms.append(ctx.last_kernel_ms())
        # hit-list exchange: ONE all_gather of K records per rank over RCCL, then the same device merge on every rank
        # (hhv_merge_hits; with one rank it merges the rank's own list, so that a step is the same job at every N)
        src = topk_buf
        if world > 1:
            dist.all_gather_into_tensor(gathered, topk_buf.cpu() if gloo else topk_buf)
            src = gathered.to(device) if gloo else gathered
            torch.cuda.current_stream().synchronize()   # the gathered records are visible to the library's stream
        _, nm = ctx.merge_hits(src.data_ptr(), world * K, K, d_out=merged_buf.data_ptr(), fetch=False)
        return merged_buf[:nm]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.sync()

    merged = None
    for _ in range(args.warmup):
        merged = step()
    kernel_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        merged = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tcells = torch.tensor([cells_per_rank], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(tcells, op=dist.ReduceOp.SUM)
        total_cells = float(tcells.item())
    else:
        total_cells = float(cells_per_rank)
    if args.dump_topk and rank == 0 and merged is not None:
        np.save(args.dump_topk, merged.cpu().numpy()[:, [shard.COL_INDEX, 0, 6, 7]])   # global id, score bits, i2, j2

    value = total_cells * args.steps / dt
    k_ms = float(np.mean(kernel_ms))
    algo_bytes = (int(rec_off[-1]) + 1) * REC_BYTES + n_local * 16 + 64 * ((Lq + 63) // 64) * REC_BYTES
    if bt:
        algo_bytes += cells_per_rank  # 1 backtrace byte per cell
    achieved_gbs = algo_bytes / (k_ms * 1e-3) / 1e9
    kernel_cells_s = cells_per_rank / (k_ms * 1e-3)

    # HBM traffic and VALU instruction count of the dominant kernel from the committed rocprofv3 PMC profile
    # (profiles/, collected with tools/profile.sh on this same command: separate --pmc passes, FETCH_SIZE doubled as
    # the guide prescribes)
    traffic = None
    valu_wave_instr = None
    headline = n_local == 100000 and Lq == 300 and Lt == 300 and not bt and args.lengths == "fixed" and not args.local
    try:
        with open(os.path.join(ROOT, "profiles", PROFILE_JSON)) as f:
            prof = json.load(f)
        if headline:
            traffic = prof.get("traffic_bytes_per_launch")
            valu_wave_instr = prof.get("valu_wave_instr_per_launch")
    except Exception:
        pass

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
        "templates_per_s": n_local * world * args.steps / dt,
        "config": {
            "workload": "Lq%d_vs_%dx_Lt%s_%s_%s" % (Lq, n_local * world, Lt if args.lengths == "fixed" else "zipf50-1000",
                                                     "local" if args.local else "global",
                                                     "backtrace_hits_top%d" % K if bt else "score_only_top%d" % K),
            "templates_per_gpu": n_local, "Lq": Lq, "Lt": Lt, "topk": K,
            "parallelism": "template-db-shard x%d, one %s all_gather of top-K" % (world, "RCCL" if backend == "nccl" else backend)
                           if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/%s (rocprofv3 PMC, bytes per launch)" % PROFILE_JSON if traffic else None,
            "kernel": "hhv_stream_kernel", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo_bytes,
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
    if world > 1 and ndev < world:
        out["config"]["oversubscribed"] = "%d ranks on %d device(s): correctness path, not a measurement" % (world, ndev)
    if valu_wave_instr:
        # executed VALU instructions from the SQ counters: issue slots used / issue slots available in the kernel's time
        lane_ops = valu_wave_instr * 64.0
        out["roofline_valu"]["counters"] = {
            "valu_wave_instr_per_launch": valu_wave_instr,
            "valu_lane_instr_per_cell": lane_ops / cells_per_rank,
            "achieved_lane_ops_per_s": lane_ops / (k_ms * 1e-3),
            "frac_of_issue_peak": lane_ops / (k_ms * 1e-3) / VALU_PEAK_LANEOPS,
            "source": "profiles/%s (SQ_INSTS_VALU per launch) / kernel_ms of this run" % PROFILE_JSON,
            # what a SIMD of this chip actually issues: a plain v_add_f32 stream in a kernel of this shape (64-thread
            # workgroups, 248 VGPRs, two waves per SIMD) retires one wave-instruction per 2.23-2.29 clk of the nominal 2.4 GHz
            # (a lone wave one per 4.5), profiles/r2_shape_ubench.txt - the ceiling the kernel's 2 waves per SIMD can reach
            "measured_issue_ceiling_lane_ops_per_s": MEASURED_ISSUE_CEILING,
            "frac_of_measured_issue_ceiling": lane_ops / (k_ms * 1e-3) / MEASURED_ISSUE_CEILING,
        }

    single = rank == 0 and world == 1 and args.virtual_shards == 1
    plain = not bt and args.lengths == "fixed"
    if single and not args.no_configs1 and n >= 10000 and plain:
        # BASELINE configs[1]: the same query vs the first 10k templates of the resident stream
        ts10 = ctx.adopt_device_stream(np.full(10000, Lt, dtype=np.int32), rec.data_ptr())
        for _ in range(2):
            ctx.align_async(ts10)
        ctx.sync()
        t1 = time.perf_counter()
        reps = 5
        ms10 = []
        for _ in range(reps):
            ctx.align_async(ts10)
            ms10.append(ctx.last_kernel_ms())
        ctx.sync()
        d10 = time.perf_counter() - t1
        out["configs1_10k_templates"] = {"cells_per_s": 10000 * Lq * Lt * reps / d10, "kernel_ms": float(np.mean(ms10))}
        ts10.free()

    if single and not args.no_cpu_baseline:   # the contract: rank 0 at N = 1 only
        out["cpu_baseline"] = cpu_baseline(args, rec, rec_off, Ls, ctx, ts, qf, qtr, n, Lq)

    if single and not args.no_configs2 and n >= 10000 and plain:
        out["configs2_backtrace_top500"] = configs2(args, torch, ctx, ts, rec, rec_off, Ls, qf, qtr, Lq, Lt, K)

    if single and not args.no_configs4 and plain:
        out["configs4_zipf"] = configs4(args, torch, capi, synth, device, dev_index, qf, qtr, Lq, K)

    if single and not args.no_next_rows and plain:
        out["next_rows"] = next_rows()

    if single and os.environ.get("HHV_DEBUG_CLK"):
        # measurement builds (-DHHV_EXP_TIMING, tools/gpu_round2i.sh) export the shader-clock totals of wave 0
        import ctypes
        lib = capi.load()
        if hasattr(lib, "hhv_debug_clk"):
            ctx.align(ts)
            buf = (ctypes.c_ulonglong * 8)()
            lib.hhv_debug_clk(buf)
            out["debug_clk"] = [int(x) for x in buf]
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(out))
    ts.free()
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def time_bt_steps(ctx, ts, K, reps=5, warm=2):
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
    out = {}
    n10 = 10000
    ts10 = ctx.adopt_device_stream(np.full(n10, Lt, dtype=np.int32), rec.data_ptr())
    sec, kms = time_bt_steps(ctx, ts10, K)
    out["10k"] = {"cells_per_s": n10 * Lq * Lt / sec, "ms_per_step": sec * 1e3, "dp_kernel_ms": kms}
    try:
        cores = os.cpu_count() or 1
        par = pyoracle.make_params(local=args.local)
        use_ref = pyoracle.have_ref()
        eng = pyoracle.Ref() if use_ref else pyoracle.Oracle()
        m = n10 if use_ref else 512
        host = rec[: int(rec_off[m])].cpu().numpy()
        tps, ttrs = unpack_templates(host, rec_off, Ls, m)
        chk = check_hits_against_cpu(ctx, ts10, eng, par, qf, qtr, tps, ttrs, cores, K, replicate=False)
        chk["cpu_kind"] = "reference" if use_ref else "port"
        out["10k"]["gpu_matches_cpu_on_sample"] = chk
    except Exception as e:  # the headline line must not depend on the side measurements
        out["10k"]["gpu_matches_cpu_on_sample"] = {"error": repr(e)}
    ts10.free()
    try:
        sec, kms = time_bt_steps(ctx, ts, K, reps=3, warm=1)
        out["%dk" % (ts.n // 1000)] = {"cells_per_s": ts.cells() / sec, "ms_per_step": sec * 1e3, "dp_kernel_ms": kms,
                                        "backtrace_bytes_written_per_launch": int(ts.records()) * 512}
    except Exception as e:
        out["%dk" % (ts.n // 1000)] = {"error": repr(e)}
    return out


def configs4(args, torch, capi, synth, device, dev_index, qf, qtr, Lq, K):
    """BASELINE configs[4] on one GPU: mixed template lengths L_t = 49 + k, k ~ Zipf(1.2) truncated to 50..1000, local mode
    (global + mixed SIMD batches hits the reference's batch-composition quirk), score-only + top-K;
    a sample is compared with the oracle (single-length batches)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    nz = 200000
    Lz = synth.zipf_lengths(0x21F, nz).astype(np.int32)
    recz, offz, Lz = gen_stream(torch, device, [Lz], [0x5EED4000], synth.PB)
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
        tps, ttrs = unpack_templates(host, offz, Lz, m)
        par = pyoracle.make_params(local=1)
        r = pyoracle.Oracle().bench_align(par, qf, qtr, tps, ttrs, threads=os.cpu_count() or 1)
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
        r = bench_prefilter.run(200000, 300, 1000)
        out["N3_prefilter"] = {"db_sequences": r["n_db"], "db_residues": r["residues"], "Lq": r["Lq"],
                               "gapless_cells_per_s": r["ungapped"]["cells_per_s"], "gapless_kernel_ms": r["ungapped"]["kernel_ms"],
                               "sw_cells_per_s": r["gapped"]["cells_per_s"], "sw_kernel_ms": r["gapped"]["kernel_ms"],
                               "mismatches_vs_oracle": r["ungapped"]["mismatches"] + r["gapped"]["mismatches"],
                               "checked": r["ungapped"]["checked"] + r["gapped"]["checked"]}
    except Exception as e:  # the headline line must not depend on the side measurements
        out["N3_prefilter"] = {"error": repr(e)}
    try:
        import bench_mac
        r = bench_mac.run(500, 300, 300, 100)
        out["N4_mac_realign"] = {"hits": r["n_hits"], "Lq": r["Lq"], "Lt": r["Lt"], "kernels_ms": r["gpu_kernels_ms"],
                                 "end_to_end_ms_host_staged_profiles": r["gpu_wall_ms_incl_host_masks"],
                                 "end_to_end_ms_resident_set": r["resident_set"]["runner_ms"],
                                 "resident_identical_to_staged": r["resident_set"]["identical_to_staged"],
                                 "hits_per_s": r["gpu_hits_per_s"],
                                 "reference_hits_per_s_1core": r.get("ref_cpu_hits_per_s_1core"),
                                 "mismatches_vs_reference": r.get("mismatches_vs_reference"), "checked": r.get("checked")}
    except Exception as e:
        out["N4_mac_realign"] = {"error": repr(e)}
    try:
        # the boundary itself: ViterbiRunner::alignment of the reference (its own translation unit, all host cores it asks
        # for) against the drop-in translation unit, wall time of the call, hits compared (tools/bench_dropin.py)
        if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libhhref_dropin.so")):
            import bench_dropin
            r = bench_dropin.run(4000, 32, 300, altalis=(1,))
            a = r["altali1"]
            out["dropin_ViterbiRunner_alignment"] = {"templates": r["n_templates"], "Lq": r["L"], "Lt": r["L"], "host_threads": r["threads"],
                                                     "reference_s": a["reference_s"], "dropin_cold_cache_s": a["dropin_cold_cache_s"],
                                                     "dropin_warm_cache_s": a["dropin_warm_cache_s"], "hits_identical": a["hits_identical"]}
    except Exception as e:
        out["dropin_ViterbiRunner_alignment"] = {"error": repr(e)}
    return out


def cpu_baseline(args, rec, rec_off, Ls, ctx, ts, qf, qtr, n, Lq):
    """The reference's own batch loop (Viterbi::Align, 8 AVX2 lanes per call, OpenMP over batches as in
    src/hhviterbirunner.cpp:122) on a bounded sample of the SAME templates, on this box's host cores.
    Also cross-checks the GPU results of the sample against it (bit-exact endpoints, equal scores)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    cores = os.cpu_count() or 1
    par = pyoracle.make_params(local=args.local)
    use_ref = pyoracle.have_ref()
    eng = pyoracle.Ref() if use_ref else pyoracle.Oracle()
    # size the sample from a short probe
    Lmean = float(np.mean(Ls))
    if args.lengths != "fixed" and not args.local and use_ref:
        # global mode + mixed-length SIMD batches hits the reference's batch-composition quirk;
        # the parity definition there is the single-length batch = the restatement
        eng, use_ref = pyoracle.Oracle(), False
    probe = min(n, 1024)
    host = rec[: int(rec_off[probe])].cpu().numpy()
    tps, ttrs = unpack_templates(host, rec_off, Ls, probe)
    r = eng.bench_align(par, qf, qtr, tps, ttrs, threads=cores)
    sec = r[0]
    rate = probe * Lq * Lmean / max(sec, 1e-9)
    sample = int(min(n, max(probe, args.cpu_seconds * rate / (Lq * Lmean))))
    sample = max(8, sample - sample % 8)
    host = rec[: int(rec_off[sample])].cpu().numpy()
    tps, ttrs = unpack_templates(host, rec_off, Ls, sample)
    r = eng.bench_align(par, qf, qtr, tps, ttrs, threads=cores)
    sec, score, i2, j2 = r[0], r[-3], r[-2], r[-1]
    n1 = max(8, min(sample, 512))
    r1 = eng.bench_align(par, qf, qtr, tps[:n1], ttrs[:n1], threads=1)
    sample_cells = float(Lq) * float(np.sum(Ls[:sample]))
    gpu = ctx.align(ts)
    ok_idx = bool(np.array_equal(gpu["i2"][:sample], i2) and np.array_equal(gpu["j2"][:sample], j2))
    ok_score = bool(np.all(gpu["score"][:sample] == score))
    maxdiff = float(np.max(np.abs(gpu["score"][:sample].astype(np.float64) - score.astype(np.float64))))
    return {
        "value": sample_cells / sec, "unit": "cells/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": "first %d templates of the benchmark set (Lq=%d, mean Lt=%.0f), %s, "
                  "OpenMP dynamic over batches, %d threads, %.2f s" % (
                      sample, Lq, Lmean, "Viterbi::Align AVX2 8 lanes/call" if use_ref else "scalar C restatement",
                      cores, sec),
        "single_thread_cells_per_s": float(Lq) * float(np.sum(Ls[:n1])) / r1[0],
        "gpu_matches_cpu_on_sample": {"endpoints_bit_exact": ok_idx, "scores_equal": ok_score, "max_abs_score_diff": maxdiff},
    }


if __name__ == "__main__":
    main()
