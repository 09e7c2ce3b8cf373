"""The merge half of the sharded top-K on the device (hhv_tset_set_global_ids, hhv_merge_hits) against the torch/numpy
statement of the same order (pyhhv/shard.py::merge_records: score descending, ties by the smaller global id, padding last)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _records(rng, m, n_pad, ties=True):
    from pyhhv import capi
    rec = np.zeros(m, dtype=capi.HIT_DTYPE)
    score = rng.normal(40.0, 25.0, m).astype(np.float32)
    if ties:
        score[rng.integers(0, m, m // 3)] = np.float32(17.25)      # many equal scores: the id decides
        score[rng.integers(0, m, m // 10)] = np.float32(-3.5)
    rec["score"] = score
    rec["viterbi_score"] = score + 1
    rec["index"] = rng.permutation(4 * m)[:m].astype(np.int32)     # distinct global ids
    rec["i2"] = rng.integers(1, 300, m)
    rec["j2"] = rng.integers(1, 300, m)
    pad = rng.choice(m, n_pad, replace=False)
    raw = rec.view(np.uint8).reshape(m, -1)
    raw[pad] = 0xFF                                                 # what hhv_topk writes beyond its n_out
    return rec


def _expected(rec, k):
    ok = rec["index"] >= 0
    v = rec[ok]
    order = np.lexsort((v["index"], -v["score"].astype(np.float64)))
    return v[order[:k]]


@pytest.mark.parametrize("m,k,n_pad", [(500, 500, 0), (500, 500, 37), (4000, 500, 400), (4096, 1000, 0), (10000, 500, 1200),
                                      (1, 5, 0), (3, 2, 3), (6000, 6000, 100),
                                      # at most 1024 records: the one-key-per-thread kernel (merge_hits_small_kernel)
                                      (64, 64, 0), (65, 10, 3), (700, 500, 0), (1000, 700, 50), (1024, 1500, 10), (1025, 500, 0)])
def test_merge_hits_equals_reference_order(m, k, n_pad):
    import torch
    from pyhhv import capi
    rng = np.random.default_rng(m * 31 + k)
    rec = _records(rng, m, n_pad)
    d = torch.from_numpy(rec.view(np.int32).reshape(m, -1).copy()).cuda()
    c = capi.Context()
    out, n = c.merge_hits(d.data_ptr(), m, k)
    exp = _expected(rec, k)
    assert n == len(exp)
    assert out.tobytes() == exp.tobytes()
    # device output buffer: the same records, 0xFF padding behind them
    dbuf = torch.zeros((k, 10), dtype=torch.int32, device="cuda")
    _, n2 = c.merge_hits(d.data_ptr(), m, k, d_out=dbuf.data_ptr(), fetch=False)
    h = dbuf.cpu().numpy()
    assert n2 == n and h[:n].tobytes() == exp.tobytes()
    assert (h[n:].view(np.uint8) == 0xFF).all()
    c.close()


def test_merge_equals_torch_merge_records():
    """the CPU/gloo tests use shard.merge_records; both statements of the order must agree"""
    import torch
    from pyhhv import capi, shard
    rng = np.random.default_rng(5)
    rec = _records(rng, 3000, 211)
    t = torch.from_numpy(rec.view(np.int32).reshape(3000, -1).copy())
    ref = shard.merge_records(torch, t, 500).numpy()
    c = capi.Context()
    out, n = c.merge_hits(t.cuda().data_ptr(), 3000, 500)
    assert n == 500 and out.view(np.int32).reshape(n, -1).tobytes() == ref.tobytes()
    c.close()


def test_topk_reports_global_ids():
    import torch  # noqa: F401
    from pyhhv import capi, synth
    rng = np.random.default_rng(11)
    q, qtr = synth.make_query(901, 120)
    tps, ttrs = [], []
    for k in range(300):
        p, tr = synth.make_template(5000 + k, int(rng.integers(40, 200)))
        tps.append(p)
        ttrs.append(tr)
    c = capi.Context(local=1)
    c.set_query(q, qtr)
    ts = c.upload(tps, ttrs)
    c.align(ts)
    base, _ = c.topk(ts, 50, raw=True)
    gids = (np.arange(300, dtype=np.int32) * 7 + 1000)[::-1].copy()
    c.set_global_ids(ts, gids)
    glob, _ = c.topk(ts, 50, raw=True)
    assert np.array_equal(glob["index"], gids[base["index"]])
    assert np.array_equal(glob["score"], base["score"])
    c.set_global_ids(ts, None)
    again, _ = c.topk(ts, 50, raw=True)
    assert again.tobytes() == base.tobytes()
    with pytest.raises(capi.HhvError):
        c.set_global_ids(ts, -np.ones(300, dtype=np.int32))
    c.close()


@pytest.mark.parametrize("n,ks", [(4096, (1, 500)), (4097, (1, 7, 500, 1024)), (20000, (500, 1024, 1025, 3000)),
                                  (70000, (1, 64, 500)), (300000, (500, 1024)),
                                  # one launch for n <= 16 384, K <= 1024 (topk_small_kernel): the candidate bound (K 1 .. 600) and
                                  # its radix branch (K near 1024: more than 1024 keys above the bound), few keys per thread
                                  (10000, (1, 2, 63, 64, 65, 500, 600, 1000, 1024)), (16384, (500, 900, 1024)), (16385, (500,)),
                                  (1500, (1, 1000, 1024, 1400)), (100, (1, 64, 100)), (1024, (1024,))])
def test_topk_selection_equals_a_full_sort(n, ks):
    """hhv_topk selects (chunks of 16 384 keys radix-selected in registers, levels until 4096 keys are left, one bitonic
    sort) instead of sorting everything: every path - final sort only (n <= 4096), one level, two levels (300 000 x 1024),
    the full-sort path for K > 1024 - against numpy's order: score descending, ties by the smaller index
    (src/hhhit.h:116-126).  The set holds only 40 distinct templates, so almost every score is shared by thousands of
    templates and the K-th key is decided by the index bytes."""
    from pyhhv import capi, synth
    q, qtr = synth.make_query(902, 6)
    base = [synth.make_template(6100 + k, 3 + k % 5) for k in range(40)]
    rng = np.random.default_rng(n)
    pick = rng.integers(0, 40, size=n)
    tps, ttrs = [base[p][0] for p in pick], [base[p][1] for p in pick]
    c = capi.Context(local=1)
    c.set_query(q, qtr)
    ts = c.upload(tps, ttrs)
    res = c.align(ts)
    score = np.asarray(res["score"], dtype=np.float32)
    order = np.lexsort((np.arange(n), -score.astype(np.float64)))
    for k in ks:
        top, nv = c.topk(ts, k, raw=True)
        assert nv == k
        assert np.array_equal(top["index"], order[:k]), (n, k)
        assert np.array_equal(top["score"], score[order[:k]])
    # fewer templates than K: all of them, the rest padding
    ts.free()
    c.close()


@pytest.mark.parametrize("mode", ["0", "2"])
def test_topk_small_set_switches(mode):
    """HHV_TOPK_SMALL = 0 (the multi-launch path of round 4) and = 2 (the one-launch kernel forced through its radix branch)
    give the records of the default path: the selection test above in a process of its own, the switch is read once."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HHV_TOPK_SMALL=mode)
    sel = "tests/test_gpu_merge.py::test_topk_selection_equals_a_full_sort"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", sel, "tests/test_gpu_merge.py::test_merge_hits_equals_reference_order",
                        "-k", "10000 or 1500 or 4096 or 700 or 500"],
                       cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
