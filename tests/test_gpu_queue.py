"""The stream kernel's work queue (-m gpu): every systolic array (one per wave, or two / four for short queries) draws stream
SEGMENTS (whole templates, >= 128 records, longest first) instead of walking one fixed range (hhv_stream_kernel.h WorkQueue, hhv_api.cpp ensure_segments).

What can go wrong there and nowhere else: a ring chunk that holds a junction between two segments (records of two places of
the stream in one 32-record chunk), the header of a segment's first template finalizing the last template of ANOTHER segment,
the terminal header behind the last segment a wave drew, the backtrace entry address of a lane whose record lies behind a
junction while its neighbour's lies in front of it, segments of exactly 128 records (a junction every fourth chunk), a stream
shorter than one chunk, a remainder of < 128 records joining the segment in front of it, fewer segments than arrays (a wave
whose second array has nothing to do).
Every case: score-only, backtrace (bytes of a sample, Hit scores and step counts of all) and a masked round, local and
global, against the oracle - the reference's definition of the path is per template, so which wave aligned it must not show."""
import numpy as np
import pytest

from pyoracle import make_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hhv():
    from pyhhv import capi
    capi.load()
    return capi


PATTERNS = {
    # every template its own segment of exactly 128 records
    "L127": ((127,), 5000),
    # pairs / fours of short templates merged into one segment, long ones of their own, lengths around the 128-record limit
    "mixed": ((63, 63, 500, 31, 31, 31, 31, 1000, 127, 128, 129, 96, 32, 1, 2, 300), 6000),
    # more waves than segments
    "few": ((300, 64, 200), 100),
    # one segment, shorter than a ring chunk
    "one_short": ((20,), 1),
    # one segment of 93 records (two templates)
    "two_short": ((60, 30), 2),
    # the remainder (31 records) joins the last segment
    "remainder": ((200, 200, 30), 3),
}


def base_templates(qf, Lq, lengths, seed):
    from pyhhv import synth
    out = {}
    for L in sorted(set(lengths)):
        v = []
        for k in range(3):
            seed += 1
            if k < 2 and L >= 2:
                v.append(synth.make_homolog(seed, qf, L=L, start=1 + (37 * k + L) % max(1, Lq - min(L, Lq) + 1)))
            else:
                v.append(synth.make_template(seed, L))
        out[L] = v
    return out


@pytest.mark.parametrize("local", [0, 1])
@pytest.mark.parametrize("name", list(PATTERNS))
@pytest.mark.parametrize("Lq", [300, 161, 150, 70])   # five / three rows per lane on 64 lanes; two 32-lane arrays; four 16-lane arrays
def test_queue_junctions(hhv, oracle, Lq, name, local):
    from pyhhv import synth
    if Lq == 161 and name not in ("mixed", "L127"):
        pytest.skip("the small cases run at the headline query length and on the short-query arrays")
    cycle, n = PATTERNS[name]
    rng = np.random.default_rng(len(name) * 100 + Lq + local)
    par = make_params(local=local)
    qf, qtr = synth.make_query(52000 + Lq, Lq)
    base = base_templates(qf, Lq, cycle, 53000 + Lq)
    Ls = [cycle[k % len(cycle)] for k in range(n)]
    var = rng.integers(0, 3, n)
    tps = [base[L][v][0] for L, v in zip(Ls, var)]
    ttrs = [base[L][v][1] for L, v in zip(Ls, var)]
    want = {(L, v): oracle.align(par, qf, qtr, base[L][v][0], base[L][v][1], want_path=True) for L in base for v in range(3)}
    w = [want[(L, v)] for L, v in zip(Ls, var)]
    w_score = np.array([a.score for a in w], dtype=np.float32)
    w_i2 = np.array([a.i2 for a in w], dtype=np.int32)
    w_j2 = np.array([a.j2 for a in w], dtype=np.int32)

    c = hhv.Context(local=par["local"], egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"],
                    ssw=par["ssw"], ss_mode=par["ss_mode"])
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)

    def check(res, what, skip=()):
        bad = [e for e in np.nonzero((res["i2"] != w_i2) | (res["j2"] != w_j2) | (res["score"] != w_score))[0] if e not in skip]
        assert not bad, (what, name, Lq, local, len(bad), bad[:8])

    for rep in range(2):    # (the ticket counter is set anew for every launch)
        check(c.align(ts), "score-only %d" % rep)
    check(c.align(ts, backtrace=True), "backtrace")
    hits = c.hits(ts)
    assert np.array_equal(hits["nsteps"], np.array([a.nsteps for a in w], dtype=np.int32))
    assert np.all(hits["score"] == np.array([a.hit_score for a in w], dtype=np.float32))
    sample = sorted(set(list(range(min(n, 24))) + list(range(max(0, n - 24), n)) + [int(e) for e in rng.integers(0, n, 24)]))
    for e in sample:
        a = w[e]
        assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (name, Lq, local, e)
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert ns == a.nsteps and np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(j_s[1:ns + 1], a.j_steps[1:ns + 1])
    # masked round (the cell-off variants read the entry the backtrace variants write): the first alignment of a few templates
    # switched off, a random mask for a few more; the others must come out as before
    res = c.align(ts, backtrace=True)
    masks = {}
    picks = sorted(set(int(e) for e in rng.integers(0, n, 12)))
    for k, e in enumerate(picks):
        a = w[e]
        if k % 2 == 0:
            masks[e] = oracle.exclude_alignment(Lq, Ls[e], a.i_steps, a.j_steps, a.nsteps)
        else:
            masks[e] = (rng.random((Lq + 1, Ls[e] + 1)) < 0.3).astype(np.uint8)
        c.set_celloff(ts, e, masks[e])
    res2 = c.align(ts, celloff=True)
    for e, m in masks.items():
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=m, want_path=True)
        assert (a.i2, a.j2) == (res2["i2"][e], res2["j2"][e]) and np.float32(a.score) == res2["score"][e], (name, Lq, local, e)
        assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:] & 0x7F, a.bt[1:, 1:] & 0x7F), (name, Lq, local, e)
    others = np.array([e not in masks for e in range(n)])
    assert np.array_equal(res2[others].view(np.uint8), res[others].view(np.uint8))
    ts.free()
    c.close()
