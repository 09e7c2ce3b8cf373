"""Queries of two strips (321 .. 640 rows) in ONE launch of two-wave workgroups (hhv_stream_kernel.h PairLds, hhv_pair_kernel):
the first strip's bottom row reaches the second through an LDS FIFO, the second wave follows the first wave's segment draws.
Against the oracle on samples, and bit for bit against the two-launch path (hhv_set_launch_policy pair_mode 0) on sets large enough that every
resident workgroup walks many segments (thousands of templates, 1-3-column templates back to back, streams shorter than the
pipeline's lag, more workgroups than segments)."""
import numpy as np
import pytest

from common import same_float, workload
from pyoracle import make_params

pytestmark = pytest.mark.gpu


def both_ways(c, fn):
    """fn() with the pair kernels and with one launch per strip (hhv_set_launch_policy of context c)"""
    out = []
    for v in (1, 0):
        c.set_launch_policy(pair_mode=v)
        try:
            out.append(fn())
        finally:
            c.set_launch_policy(pair_mode=-1)
    return out


# 321 .. 640: one pair launch; beyond: chains of launches (three strips = pair + strip, four = two pairs, five = two pairs +
# strip, seven = three pairs + strip; five-row strips of local mode and of backtrace plans stay launches of their own)
@pytest.mark.parametrize("Lq", [321, 384, 431, 448, 449, 512, 513, 576, 600, 640, 641, 700, 768, 960, 1000, 1280, 1300, 1600, 2000])
@pytest.mark.parametrize("local", [0, 1])
def test_pair_equals_oracle_and_two_launches(oracle, Lq, local):
    from pyhhv import capi
    par = make_params(local=local, egq=0.0 if Lq % 2 else 0.2, egt=0.0 if Lq % 2 else 0.1)
    n = 24
    qf, qtr, tps, ttrs = workload(300 + Lq, Lq, n, 1, 420)
    c = capi.Context(local=local, egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"])
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)

    def run():
        so = c.align(ts).copy()
        res = c.align(ts, backtrace=True).copy()
        hits = c.hits(ts).copy()
        mats = [c.backtrace_matrix(ts, e) for e in (0, 5, 11)]
        return so, res, hits, mats

    (so1, res1, hits1, mats1), (so0, res0, hits0, mats0) = both_ways(c, run)
    for e in range(n):  # (first against the oracle, template by template: a failure names the template and the path)
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=False)
        for tag, r in (("two launches", so0), ("pair", so1), ("two launches bt", res0), ("pair bt", res1)):
            assert (a.i2, a.j2) == (r["i2"][e], r["j2"][e]) and same_float(a.score, r["score"][e]), (tag, Lq, e, tps[e].shape[0] - 1, a.score, r["score"][e], (a.i2, a.j2), (r["i2"][e], r["j2"][e]))
    assert so1.tobytes() == so0.tobytes() and res1.tobytes() == res0.tobytes() and hits1.tobytes() == hits0.tobytes()
    for a, b in zip(mats1, mats0):
        assert np.array_equal(a, b)
    for e in range(n):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_path=True)
        assert (a.i2, a.j2) == (res1["i2"][e], res1["j2"][e]) == (so1["i2"][e], so1["j2"][e]), (Lq, e)
        assert same_float(a.score, res1["score"][e]) and same_float(a.score, so1["score"][e])
        assert hits1["nsteps"][e] == a.nsteps and same_float(hits1["score"][e], a.hit_score)
    for k, e in enumerate((0, 5, 11)):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=True)
        assert np.array_equal(mats1[k][1:, 1:], a.bt[1:, 1:]), (Lq, e)
    ts.free()
    c.close()


@pytest.mark.parametrize("Lq,local", [(431, 0), (431, 1), (512, 1), (640, 0), (700, 1), (1000, 0), (1000, 1), (1300, 0)])
def test_pair_on_large_sets_equals_two_launches(Lq, local):
    """every resident workgroup walks tens of segments: junctions, the segment FIFO, the carry FIFO's wrap-around and both
    flow-control waits; tiny templates back to back (headers in consecutive steps: the best travels through the FIFO too)"""
    from pyhhv import capi, synth
    rng = np.random.default_rng(Lq + local)
    qf, qtr = synth.make_query(7000 + Lq, Lq)
    base = []
    for k in range(60):
        Lt = [1, 1, 2, 3, 1, 300, 64, 129, 130, 511][k % 10] + (k // 10 if k % 10 >= 5 else 0)
        base.append(synth.make_homolog(8000 + k, qf, L=Lt) if k % 3 == 0 else synth.make_template(8000 + k, Lt))
    for n in (40000, 700, 3):
        pick = rng.integers(0, 60, size=n)
        tps, ttrs = [base[p][0] for p in pick], [base[p][1] for p in pick]
        c = capi.Context(local=local)
        c.set_query(qf, qtr)
        ts = c.upload(tps, ttrs)

        def run():
            so = c.align(ts).copy()
            res = c.align(ts, backtrace=True).copy()
            hits = c.hits(ts).copy()
            return so, res, hits

        (so1, res1, hits1), (so0, res0, hits0) = both_ways(c, run)
        assert so1.tobytes() == so0.tobytes(), (Lq, n)
        assert res1.tobytes() == res0.tobytes(), (Lq, n)
        assert hits1.tobytes() == hits0.tobytes(), (Lq, n)
        # the same template gives the same result wherever it sits in the stream
        first = {}
        for e, p in enumerate(pick):
            if p in first:
                f = first[p]
                assert so1["score"][e].tobytes() == so1["score"][f].tobytes() and so1["i2"][e] == so1["i2"][f] and so1["j2"][e] == so1["j2"][f]
            else:
                first[p] = e
        ts.free()
        c.close()
