"""Secondary-structure scoring on the GPU (Viterbi::AlignWith[Out]CellOffAndSS +
the score_ss bookkeeping of Viterbi::ScoreForBacktrace) against the oracle, which is pinned to the
reference's -DVITERBI_SS_SCORE builds in tests/test_oracle_vs_reference.py."""
import numpy as np
import pytest

from common import same_float, workload
from pyoracle import SSInfo, make_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Lq,local", [(80, 1), (300, 0), (330, 1), (600, 1), (640, 0)])   # (600 / 640: two passes of five rows per lane)
def test_ss_modes_match_oracle(oracle, Lq, local):
    from pyhhv import capi
    rng = np.random.default_rng(5 + Lq)
    S73 = rng.normal(0, 1, (8, 4, 11)).astype(np.float32)
    S33 = rng.normal(0, 1, (4, 11, 4, 11)).astype(np.float32)
    S37 = rng.normal(0, 1, (4, 11, 8)).astype(np.float32)
    par = make_params(local=local, ss_mode=2)
    n = 9
    qf, qtr, tps, ttrs = workload(9 + Lq, Lq, n, 40, 260, homolog_every=2)
    c = capi.Context(local=local, ssw=par["ssw"], ss_mode=2)
    c.set_query(qf, qtr)
    c.set_ss_tables(S73, S33, S37)
    q_ss = (rng.integers(0, 4, Lq + 1), rng.integers(0, 11, Lq + 1), rng.integers(0, 8, Lq + 1))
    c.set_query_ss(*q_ss)
    t_ss = [(rng.integers(0, 4, p.shape[0]), rng.integers(0, 11, p.shape[0]), rng.integers(0, 8, p.shape[0]))
            for p in tps]
    ts = c.upload(tps, ttrs, t_ss)
    for mode in (4, 2, 1, 0):
        c.set_ss_mode(mode)
        ss = SSInfo(mode, *q_ss, S73, S33, S37) if mode else None
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        for e in range(n):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], ss=ss, t_ss=t_ss[e] if mode else None, want_path=True)
            assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e]), (mode, e)
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:])
            assert same_float(hits["score"][e], a.hit_score) and same_float(hits["score_ss"][e], a.score_ss)
            assert hits["nsteps"][e] == a.nsteps
    # score-only SS kernel == backtrace SS kernel
    c.set_ss_mode(4)
    r1 = c.align(ts)
    r2 = c.align(ts, backtrace=True)
    assert np.array_equal(r1.view(np.uint8), r2.view(np.uint8))
    ts.free()
    c.close()
