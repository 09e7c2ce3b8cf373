"""Real HH-suite profiles: the reference's data/query.hhm, read + prepared + aligned by the reference's
own code at fixture-generation time (tests/golden/make_real_fixture.py).  The committed prepared tensors
and reference outputs must be reproduced by the oracle (CPU) and by the HIP engine (GPU; Lq = 431 runs
as two strips of the multi-pass kernel)."""
import hashlib
import json
import os

import numpy as np
import pytest

from pyoracle import make_params
from real_fixture import HERE, load_prepared, window


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def gold():
    with open(os.path.join(HERE, "golden", "query_hhm_golden.json")) as f:
        return json.load(f)


def bits(x):
    return int(np.float32(x).view(np.uint32))


def test_fixture_sanity():
    qp, qtr, tp, ttr = load_prepared()
    assert qp.shape == (432, 20) and qtr.shape == (432, 7) and tp.shape == (432, 20)
    assert np.allclose(qp[1:].sum(axis=1), 1.0, atol=1e-3)          # query columns are probabilities
    assert np.all(qtr[1:-1, 0] <= 0) and qtr[0, 2] < -1000           # log2 transitions; no M->D out of column 0


def test_oracle_reproduces_reference_on_real_profile(oracle):
    qp, qtr, tp, ttr = load_prepared()
    for g in gold()["results"]:
        par = make_params(local=g["local"])
        p, tr = window(tp, ttr, *g["window"])
        a = oracle.align(par, qp, qtr, p, tr, want_path=True)
        ns = a.nsteps
        assert bits(a.score) == g["score_bits"] and (a.i2, a.j2) == (g["i2"], g["j2"])
        assert sha(a.bt[1:, 1:]) == g["bt_sha256"] and (ns, a.matched_cols) == (g["nsteps"], g["matched_cols"])
        assert sha(a.i_steps[1:ns + 1], a.j_steps[1:ns + 1], a.states[1:ns + 1]) == g["path_sha256"]
        assert sha(a.S[1:ns + 1]) == g["S_sha256"] and bits(a.hit_score) == g["hit_score_bits"]


@pytest.mark.gpu
def test_gpu_reproduces_reference_on_real_profile():
    from pyhhv import capi
    qp, qtr, tp, ttr = load_prepared()
    G = gold()["results"]
    for local in (1, 0):
        gs = [g for g in G if g["local"] == local]
        wins = [window(tp, ttr, *g["window"]) for g in gs]
        c = capi.Context(local=local)
        c.set_query(qp, qtr)
        ts = c.upload([w[0] for w in wins], [w[1] for w in wins])
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        for e, g in enumerate(gs):
            assert bits(res["score"][e]) == g["score_bits"] and (res["i2"][e], res["j2"][e]) == (g["i2"], g["j2"])
            assert sha(c.backtrace_matrix(ts, e)[1:, 1:]) == g["bt_sha256"]
            ns, i_s, j_s, st, S = c.hit_path(ts, e)
            assert (ns, hits["matched_cols"][e]) == (g["nsteps"], g["matched_cols"])
            assert sha(i_s[1:ns + 1], j_s[1:ns + 1], st[1:ns + 1]) == g["path_sha256"]
            assert sha(S[1:ns + 1]) == g["S_sha256"] and bits(hits["score"][e]) == g["hit_score_bits"]
            assert abs(float(hits["score"][e]) - g["hit_score"]) <= 1e-4
        ts.free()
        c.close()
