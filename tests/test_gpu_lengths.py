"""GPU parity at the reference's real length limits (run with -m gpu).

The reference truncates every HMM to par.maxres - 1 = 20 000 columns (maxres = 20001: src/hhdecl.cpp:11, the truncation
rule src/hhblits.cpp:1186-1196 and HMM::Read), so 20 001 x 20 001 is the largest matrix a search can ever ask for.  The
library itself claims Lq <= 32 767 and Lt <= 65 535 (include/hhviterbi_hip.h): 63-pass carries, the 16-bit column index in
the record meta word, the (i2 << 16) | j2 packing of the best cell, path pools of Lq + Lt + 2 entries and a backtrace
buffer of ~1.7 B/cell are exercised here against the oracle - score bits, end points, every path step, the per-step S and
the Hit score; one masked second round; local and global mode; the two limit cases and HHV_E_LIMIT just beyond them.
"""
import numpy as np
import pytest

from common import same_float
from pyoracle import make_params
from pyhhv import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hhv():
    from pyhhv import capi
    capi.load()
    return capi


def ctx_for(hhv, par):
    return hhv.Context(local=par["local"], egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"],
                       ssw=par["ssw"], ss_mode=par["ss_mode"])


def long_set(seed, Lq, Lts):
    """query + one template per entry of Lts; the first is derived from the query (a long true alignment), the rest random"""
    qf, qtr = synth.make_query(9000 + seed, Lq)
    tps, ttrs = [], []
    for k, Lt in enumerate(Lts):
        p, tr = synth.make_homolog(9100 + 17 * seed + k, qf, L=Lt) if k == 0 else synth.make_template(9200 + 17 * seed + k, Lt)
        tps.append(p)
        ttrs.append(tr)
    return qf, qtr, tps, ttrs


def check_paths(c, oracle, par, ts, qf, qtr, tps, ttrs, res, hits, celloff=None):
    for e in range(len(tps)):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=None if celloff is None else celloff[e], want_path=True)
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]), e
        assert same_float(a.score, res["score"][e]), e
        h = hits[e]
        assert h["index"] == e and h["nsteps"] == a.nsteps and h["matched_cols"] == a.matched_cols
        assert (h["i1"], h["j1"]) == (a.i_steps[a.nsteps], a.j_steps[a.nsteps])
        assert same_float(h["score"], a.hit_score), (e, h["score"], a.hit_score)
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert ns == a.nsteps
        assert np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1])
        assert np.array_equal(j_s[1:ns + 1], a.j_steps[1:ns + 1])
        assert np.array_equal(st[1:ns + 1], a.states[1:ns + 1])
        assert np.array_equal(S[1:ns + 1], a.S[1:ns + 1])


# (Lq, template lengths, local): 5 000 = 16 passes of 64 x 5 rows, 20 001 = 63 passes (the last one with a single row)
SCORE_CASES = [
    (5000, [5000, 20001, 37], 0),
    (5000, [5000, 777], 1),
    (20001, [20001, 5000, 1], 0),
    (20001, [20001, 300], 1),
]


@pytest.mark.parametrize("case", range(len(SCORE_CASES)))
def test_score_only_at_maxres(hhv, oracle, case):
    Lq, Lts, local = SCORE_CASES[case]
    par = make_params(local=local, egq=0.0 if case % 2 == 0 else 0.2, egt=0.0 if case % 2 == 0 else 0.1)
    qf, qtr, tps, ttrs = long_set(case, Lq, Lts)
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    res = c.align(ts)
    for e in range(len(tps)):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=False)
        assert res["index"][e] == e
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]), (case, e, a.i2, a.j2, res["i2"][e], res["j2"][e])
        assert same_float(a.score, res["score"][e]), (case, e, a.score, res["score"][e])
    ts.free()
    c.close()


BT_CASES = [
    (5000, [5000, 1300], 0),
    (5000, [4100, 5000], 1),
    (20001, [20001, 450], 1),
    (20001, [20001], 0),
]


@pytest.mark.parametrize("case", range(len(BT_CASES)))
def test_backtrace_at_maxres(hhv, oracle, case):
    Lq, Lts, local = BT_CASES[case]
    par = make_params(local=local)
    qf, qtr, tps, ttrs = long_set(20 + case, Lq, Lts)
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    res = c.align(ts, backtrace=True)
    hits = c.hits(ts)
    check_paths(c, oracle, par, ts, qf, qtr, tps, ttrs, res, hits)
    if Lq <= 5000:  # every backtrace byte as well (25 MB per matrix)
        for e in range(len(tps)):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=True)
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (case, e)
    ts.free()
    c.close()


@pytest.mark.parametrize("Lq,Lt,local", [(5000, 5000, 1), (20001, 20001, 0)])
def test_masked_second_round_at_maxres(hhv, oracle, Lq, Lt, local):
    """alt-alignment round 2 (src/hhviterbirunner.cpp:152-164): the first path masked by ExcludeAlignment, re-aligned"""
    par = make_params(local=local)
    qf, qtr, tps, ttrs = long_set(40 + local, Lq, [Lt, 611])
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    c.align(ts, backtrace=True)
    c.hits(ts)
    masks = []
    for e in range(len(tps)):
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        masks.append(oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, i_s, j_s, ns))
        c.set_celloff(ts, e, masks[e])
    res = c.align(ts, celloff=True)
    hits = c.hits(ts)
    check_paths(c, oracle, par, ts, qf, qtr, tps, ttrs, res, hits, celloff=masks)
    ts.free()
    c.close()


def test_masks_built_on_the_device_at_maxres(hhv, oracle):
    """hhv_set_celloff_paths: the same second round with the mask built on the device from the first round's paths"""
    par = make_params(local=1)
    Lq = 20001
    qf, qtr, tps, ttrs = long_set(47, Lq, [20001, 90])
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    c.align(ts, backtrace=True)
    c.hits(ts)
    masks, paths = [], []
    for e in range(len(tps)):
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        masks.append(oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, i_s, j_s, ns))
        paths.append((ns, i_s, j_s))
    c.set_celloff_paths(ts, [(e, p[0], p[1], p[2]) for e, p in enumerate(paths)])
    res = c.align(ts, celloff=True)
    hits = c.hits(ts)
    check_paths(c, oracle, par, ts, qf, qtr, tps, ttrs, res, hits, celloff=masks)
    ts.free()
    c.close()


def test_template_length_limit_65535(hhv, oracle):
    """Lt = 65 535: the record meta word keeps j in 16 bits, the best cell is packed as (i2 << 16) | j2"""
    for local in (0, 1):
        par = make_params(local=local)
        Lq, Lt = 300, 65535
        qf, qtr = synth.make_query(9301, Lq)
        # the query's homolog sits at the END of the long template: the best cell has j2 near 65 535
        hp, htr = synth.make_homolog(9302, qf, L=Lq)
        tp, ttr = synth.make_template(9303, Lt)
        tp[Lt - Lq + 1:Lt + 1] = hp[1:Lq + 1]
        tps, ttrs = [tp, synth.make_template(9304, 65535)[0]], [ttr, synth.make_template(9304, 65535)[1]]
        c = ctx_for(hhv, par)
        c.set_query(qf, qtr)
        ts = c.upload(tps, ttrs)
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        check_paths(c, oracle, par, ts, qf, qtr, tps, ttrs, res, hits)
        assert max(res["j2"]) > 32768  # (an end point beyond 15 bits of j2)
        so = c.align(ts)  # the score-only variant as well
        for e in range(2):
            assert (so["i2"][e], so["j2"][e]) == (res["i2"][e], res["j2"][e]) and same_float(so["score"][e], res["score"][e])
        ts.free()
        c.close()


def test_query_length_limit_32767(hhv, oracle):
    """Lq = 32 767: 103 passes; i2 takes the 15 bits above j2"""
    for local in (0, 1):
        par = make_params(local=local)
        Lq = 32767
        qf, qtr = synth.make_query(9401, Lq)
        hp, htr = synth.make_homolog(9402, qf, L=Lq)
        Lt = 400
        tp, ttr = synth.make_template(9403, Lt)
        tp[1:Lt + 1] = hp[Lq - Lt + 1:Lq + 1]  # matches the END of the query: i2 near 32 767
        tps, ttrs = [tp, synth.make_template(9404, 2000)[0]], [ttr, synth.make_template(9404, 2000)[1]]
        c = ctx_for(hhv, par)
        c.set_query(qf, qtr)
        ts = c.upload(tps, ttrs)
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        check_paths(c, oracle, par, ts, qf, qtr, tps, ttrs, res, hits)
        assert max(res["i2"]) > 16384  # (an end point that needs the 15th bit of i2)
        so = c.align(ts)
        for e in range(2):
            assert (so["i2"][e], so["j2"][e]) == (res["i2"][e], res["j2"][e]) and same_float(so["score"][e], res["score"][e])
        ts.free()
        c.close()


def test_beyond_the_limits_is_an_error_not_a_crash(hhv):
    from pyhhv import capi
    c = capi.Context(local=1)
    qf, qtr = synth.make_query(9501, 32768)
    with pytest.raises(capi.HhvError, match=r"error -5.*32767"):  # HHV_E_LIMIT
        c.set_query(qf, qtr)
    qf, qtr = synth.make_query(9502, 50)
    c.set_query(qf, qtr)
    tp, ttr = synth.make_template(9503, 65536)
    with pytest.raises(capi.HhvError, match=r"error -5.*65535"):
        c.upload([tp], [ttr])
    tp, ttr = synth.make_template(9504, 60)
    ts = c.upload([tp], [ttr])
    assert len(c.align(ts)) == 1  # the context is still usable
    ts.free()
    c.close()
