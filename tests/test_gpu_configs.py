"""BASELINE.json configs[2] and configs[4] at their sizes, and bench.py's multi-rank path on the GPU.

configs[2]: Lq = 300 vs 10 000 templates of 300 columns, global, with backtrace: EVERY template is compared with the
            reference's own batch loop (Viterbi::Align + Viterbi::Backtrace + Viterbi::ScoreForBacktrace,
            src/hhviterbi.cpp:83-160,195-281, run by oracle/_ref under OpenMP; the C restatement when _ref is absent):
            ViterbiResult, alignment start, nsteps, matched_cols, Hit.score, the whole path (i, j, state per step) and the
            per-step scores S through checksums, and the top-500 by Hit.score from the device against a host sort.
configs[4]: 20 000 templates with Zipf-distributed lengths 50..1000, local mode, every template against the oracle.
multi-rank: bench.py --gpus 2 (ranks started by bench.py itself, sharing this GPU, exchange over gloo) must merge to the
            same top-K as ONE rank holding both shards.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from pyoracle import Oracle, Ref, have_ref, make_params, path_hashes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_db(Ls, first_gid=0):
    """the templates first_gid .. first_gid + len(Ls) - 1 of the benchmark database (pyhhv/synth_stream.py: every template's
    columns come from its own splitmix64 -> xoshiro256** stream, seed 0x5EED0000 + global id)"""
    import torch
    from pyhhv import synth, synth_stream
    dev = torch.device("cuda", 0)
    rec, rec_off, Ls = synth_stream.gen_stream(torch, dev, first_gid + np.arange(len(Ls)), Ls, synth.PB)
    torch.cuda.synchronize()
    host = rec[: int(rec_off[-1])].cpu().numpy()
    tps, ttrs = synth_stream.unpack_templates(host, rec_off, Ls, len(Ls))
    return rec, rec_off, Ls, tps, ttrs


def test_device_streams_follow_the_prescribed_prng():
    """the torch generator on the GPU = the numpy restatement of splitmix64 -> xoshiro256** (known answers: tests/test_synth_stream.py)"""
    import torch
    from pyhhv import synth_stream
    gids = np.array([0, 1, 99999, 123456, 999999])
    got = synth_stream.uniforms_torch(torch, torch.device("cuda", 0), gids, 200).cpu().numpy()
    assert np.array_equal(got, synth_stream.uniforms_np(gids, 200))


def test_configs2_10k_backtrace_hits_top500_vs_reference():
    from pyhhv import capi, synth
    Lq, Lt, n, K = 300, 300, 10000, 500
    qf, qtr = synth.make_query(0x51000000, Lq)
    rec, rec_off, Ls, tps, ttrs = build_db(np.full(n, Lt, dtype=np.int32))
    par = make_params(local=0)
    eng = Ref() if have_ref() else Oracle()
    want = eng.bench_hits(par, qf, qtr, tps, ttrs, threads=os.cpu_count() or 1, replicate=False)
    c = capi.Context(local=0)
    c.set_query(qf, qtr)
    ts = c.adopt_device_stream(Ls, rec.data_ptr())
    res = c.align(ts, backtrace=True)
    hits = c.hits(ts)
    assert np.array_equal(res["score"], want["score"]) and np.array_equal(res["i2"], want["i2"]) and np.array_equal(res["j2"], want["j2"])
    for f in ("i1", "j1", "i2", "j2", "nsteps", "matched_cols"):
        assert np.array_equal(hits[f], want[f]), f
    assert np.array_equal(hits["viterbi_score"], want["score"])
    assert np.array_equal(hits["score"], want["hit_score"])          # bit-exact (north-star tolerance: 1e-4)
    off, pi, pj, pst, pS = c.hit_path_pool(ts)
    ph, sh = path_hashes(off, pi, pj, pst, pS, hits["nsteps"])
    assert np.array_equal(ph, want["path_hash"]), "paths differ"
    assert np.array_equal(sh, want["s_hash"]), "per-step scores S differ"
    top, nv = c.topk(ts, K)
    order = np.lexsort((np.arange(n), -want["hit_score"].astype(np.float64)))[:K]
    assert nv == K and np.array_equal(top["index"], order)
    assert np.array_equal(top["score"], want["hit_score"][order])
    ts.free()
    c.close()


def test_configs4_zipf_local_20k_vs_oracle():
    from pyhhv import capi, synth
    Lq, n = 300, 20000
    qf, qtr = synth.make_query(0x51000000, Lq)
    Ls = synth.zipf_lengths(0x21F, n).astype(np.int32)
    assert Ls.min() >= 50 and Ls.max() <= 1000 and len(set(Ls.tolist())) > 100
    rec, rec_off, Ls, tps, ttrs = build_db(Ls)
    par = make_params(local=1)
    sec, score, i2, j2 = Oracle().bench_align(par, qf, qtr, tps, ttrs, threads=os.cpu_count() or 1)
    c = capi.Context(local=1)
    c.set_query(qf, qtr)
    ts = c.adopt_device_stream(Ls, rec.data_ptr())
    res = c.align(ts)
    assert np.array_equal(res["i2"], i2) and np.array_equal(res["j2"], j2)
    assert np.array_equal(res["score"], score)
    # with backtrace the end points must not change, and a sample of Hit scores equals the oracle's
    res_bt = c.align(ts, backtrace=True)
    assert np.array_equal(res_bt.view(np.uint8), res.view(np.uint8))
    hits = c.hits(ts)
    m = 2000
    want = Oracle().bench_hits(par, qf, qtr, tps[:m], ttrs[:m], threads=os.cpu_count() or 1)
    for f in ("i1", "j1", "nsteps", "matched_cols"):
        assert np.array_equal(hits[f][:m], want[f]), f
    assert np.array_equal(hits["score"][:m], want["hit_score"])
    top, nv = c.topk(ts, 500, raw=True)
    order = np.lexsort((np.arange(n), -score.astype(np.float64)))[:500]
    assert nv == 500 and np.array_equal(top["index"], order)
    ts.free()
    c.close()


def run_bench(extra, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-configs1",
           "--no-configs2", "--no-configs4", "--no-next-rows", "--no-upload", "--no-fast-mode"] + extra
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("lengths,bt", [("fixed", 0), ("zipf", 1)])
def test_bench_two_ranks_merge_equals_one_rank(tmp_path, lengths, bt):
    """`python bench.py --gpus 2` as the driver invokes it (no torchrun environment): bench.py starts both ranks; they share
    this GPU, so the exchange runs over gloo.  The merged top-K must equal the top-K of one rank holding both shards."""
    common = ["--templates", "1500", "--lt", "120", "--lq", "150", "--topk", "64", "--lengths", lengths, "--backtrace", str(bt),
              "--local", "1" if lengths == "zipf" else "0"]
    two = run_bench(["--gpus", "2", "--dump-topk", str(tmp_path / "two.npy")] + common)
    one = run_bench(["--gpus", "1", "--virtual-shards", "2", "--dump-topk", str(tmp_path / "one.npy")] + common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["templates_total"] == 3000 and one["config"]["templates_total"] == 3000
    assert two["config"]["templates_per_gpu"] == 1500 and one["config"]["templates_per_gpu"] == 3000
    # ONE global database cut by hhv_shard_plan: both runs report the same plan, balanced on stream records
    assert two["config"]["shards"] == one["config"]["shards"]
    sh = two["config"]["shards"]
    assert sum(sh["templates_per_shard"]) == 3000 and sh["imbalance_max_over_mean"] < 1.05
    if lengths == "zipf":
        assert sh["templates_per_shard"][0] != sh["templates_per_shard"][1]    # LPT balances records, not template counts
    a, b = np.load(tmp_path / "two.npy"), np.load(tmp_path / "one.npy")
    assert a.shape == (64, 4) and np.array_equal(a, b)
    ids = set(a[:, 0].tolist())
    assert len(ids) == 64
    from pyhhv import capi, synth
    Lg = synth.zipf_lengths(0x21F, 3000) if lengths == "zipf" else np.full(3000, 120)
    owner = capi.shard_plan(Lg, 2)
    assert set(owner[list(ids)].tolist()) == {0, 1}                             # hits from both shards, global ids


def test_bench_one_rank_through_rccl_equals_plain_one_rank(tmp_path):
    """--force-dist: ONE rank through init_process_group("nccl") (= RCCL) + all_gather_into_tensor on device records +
    hhv_merge_hits, the streams ordered by events - the branch every multi-GPU run takes - must give the plain one-rank
    top-K.  (RCCL works with one rank; two ranks cannot share this box's single GPU.)"""
    common = ["--templates", "3000", "--lt", "120", "--lq", "150", "--topk", "64", "--steps", "3"]
    env = {"HHV_BENCH_BACKEND": "nccl"}
    a = run_bench(["--gpus", "1", "--force-dist", "--dump-topk", str(tmp_path / "dist.npy")] + common, env_extra=env)
    b = run_bench(["--gpus", "1", "--dump-topk", str(tmp_path / "plain.npy")] + common)
    assert "RCCL" in a["config"]["parallelism"] and b["config"]["parallelism"] == "single GPU"
    x, y = np.load(tmp_path / "dist.npy"), np.load(tmp_path / "plain.npy")
    assert x.shape == (64, 4) and np.array_equal(x, y)
    # and the same through the backtrace + Hit-score step
    a = run_bench(["--gpus", "1", "--force-dist", "--backtrace", "1", "--dump-topk", str(tmp_path / "dist_bt.npy")] + common, env_extra=env)
    b = run_bench(["--gpus", "1", "--backtrace", "1", "--dump-topk", str(tmp_path / "plain_bt.npy")] + common)
    x, y = np.load(tmp_path / "dist_bt.npy"), np.load(tmp_path / "plain_bt.npy")
    assert x.shape == (64, 4) and np.array_equal(x, y)
