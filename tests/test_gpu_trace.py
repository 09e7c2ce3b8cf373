"""The two forms of the backtrace walk (Viterbi::Backtrace, src/hhviterbi.cpp:83-160) against each other (-m gpu): one lane per
template (hhv_trace_kernel: large sets) and one wavefront per template (hhv_trace_wave_kernel, round 5: a round trip per RUN of
the path - the sets of real searches).  Every other GPU test runs the walk the library picks for its (small) set, i.e. the
wavefront form, against the oracle; here both are forced (hhv_set_launch_policy trace_mode) on the same backtrace buffer and must
agree byte for byte - hits, path pools, per-step scores - and a sample is compared with the oracle: long gaps (runs of gap
states), paths that end at the matrix border, local alignments that stop on a STOP code, 1-column templates, queries of one
strip, two strips (the plan is looked up per cell) and the short-query arrays, masked second rounds."""
import numpy as np
import pytest

from pyoracle import make_params

pytestmark = pytest.mark.gpu


def gapped_homolog(synth, seed, qf, qtr, L, rng):
    """a template derived from the query with blocks cut out and random blocks put in: alignments with long gap runs"""
    p, tr = synth.make_homolog(seed, qf, L=L)
    if L > 40:
        a = int(rng.integers(5, L // 2))
        w = int(rng.integers(3, min(30, L // 3)))
        r, _ = synth.make_template(seed + 1, L)
        p = p.copy()
        p[a:a + w] = r[a:a + w]                      # a stretch that does not match: an insertion / deletion pair
        if rng.random() < 0.5 and L - a - w > 10:    # and a shift of the rest: a pure gap run
            p[a + w:] = np.roll(p[a + w:], int(rng.integers(1, 8)), axis=0)
    return p, tr


@pytest.mark.parametrize("local", [0, 1])
@pytest.mark.parametrize("Lq", [300, 431, 150, 70, 1000])
def test_wave_walk_equals_lane_walk(oracle, Lq, local):
    from pyhhv import capi, synth
    rng = np.random.default_rng(91 + Lq + local)
    par = make_params(local=local, egq=0.1 if Lq == 150 else 0.0, egt=0.05 if Lq == 150 else 0.0)
    qf, qtr = synth.make_query(81000 + Lq, Lq)
    base = []
    for k, L in enumerate([1, 2, 3, 40, 64, 90, 130, 200, 257, 300, 420, 700][: 12 if Lq <= 431 else 9]):
        for v in range(3):
            if v < 2 and L >= 4:
                base.append(gapped_homolog(synth, 82000 + 10 * k + 3 * v, qf, qtr, L, rng))
            else:
                base.append(synth.make_template(83000 + 10 * k + v, L))
    n = 900
    pick = rng.integers(0, len(base), n)
    tps, ttrs = [base[p][0] for p in pick], [base[p][1] for p in pick]
    c = capi.Context(local=local, egq=par["egq"], egt=par["egt"])
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)

    def both(celloff):
        out = []
        for mode in (1, 0):
            c.set_launch_policy(trace_mode=mode)
            c.align(ts, backtrace=True, celloff=celloff)
            hits = c.hits(ts).copy()
            out.append((hits, c.hit_path_pool(ts)))
        c.set_launch_policy(trace_mode=-1)
        return out

    (h1, p1), (h0, p0) = both(False)
    assert h1.tobytes() == h0.tobytes()
    off = p1[0]
    for e in range(n):   # the pools hold garbage behind a path's last step (and at index 0): compare steps 1 .. nsteps
        ns = int(h1["nsteps"][e])
        a, b = int(off[e]) + 1, int(off[e]) + ns + 1
        for x, y in zip(p1[1:], p0[1:]):
            assert np.array_equal(x[a:b], y[a:b]), (Lq, local, e)
    # the compact records of hhv_hit_paths_packed: steps 0 .. nsteps of every path and nothing else, entry 0 all zero
    c.align(ts, backtrace=True)
    hk = c.hits(ts).copy()
    assert hk.tobytes() == h1.tobytes()
    koff, ki, kj, kst, kS = c.hit_paths_packed(ts, hk)
    assert int(koff[n]) == int(hk["nsteps"].sum()) + n
    for e in range(n):
        ns = int(hk["nsteps"][e])
        a, o = int(off[e]), int(koff[e])
        assert int(koff[e + 1]) - o == ns + 1
        assert ki[o] == 0 and kj[o] == 0 and kst[o] == 0 and kS[o] == 0.0
        assert np.array_equal(ki[o + 1:o + ns + 1].astype(np.int32), p1[1][a + 1:a + ns + 1]), (Lq, local, e)
        assert np.array_equal(kj[o + 1:o + ns + 1].astype(np.int32), p1[2][a + 1:a + ns + 1])
        assert np.array_equal(kst[o + 1:o + ns + 1], p1[3][a + 1:a + ns + 1])
        assert kS[o + 1:o + ns + 1].tobytes() == p1[4][a + 1:a + ns + 1].tobytes()
    seen = {}
    for e in range(n):
        if pick[e] in seen:
            continue
        seen[pick[e]] = e
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_path=True)
        ns = a.nsteps
        assert (h1["nsteps"][e], h1["matched_cols"][e], h1["i1"][e], h1["j1"][e]) == (ns, a.matched_cols, a.i_steps[ns], a.j_steps[ns]), (Lq, local, e)
        o = int(off[e])
        assert np.array_equal(p1[1][o + 1:o + ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(p1[2][o + 1:o + ns + 1], a.j_steps[1:ns + 1])
        assert np.array_equal(p1[3][o + 1:o + ns + 1], a.states[1:ns + 1]) and np.float32(a.hit_score) == h1["score"][e]
    # masked second round: every template's first alignment switched off (paths that start somewhere else, often at a border)
    paths = [(e, int(h1["nsteps"][e]), p1[1][off[e]: off[e] + h1["nsteps"][e] + 1], p1[2][off[e]: off[e] + h1["nsteps"][e] + 1]) for e in range(n)]
    c.set_celloff_paths(ts, paths)
    c.set_launch_policy(trace_mode=1)
    c.align(ts, celloff=True)
    g1 = c.hits(ts).copy()
    q1 = c.hit_path_pool(ts)
    c.set_celloff_paths(ts, paths)
    c.set_launch_policy(trace_mode=0)
    c.align(ts, celloff=True)
    g0 = c.hits(ts).copy()
    q0 = c.hit_path_pool(ts)
    c.set_launch_policy(trace_mode=-1)
    assert g1.tobytes() == g0.tobytes()
    for e in range(n):
        ns = int(g1["nsteps"][e])
        a, b = int(off[e]) + 1, int(off[e]) + ns + 1
        for x, y in zip(q1[1:], q0[1:]):
            assert np.array_equal(x[a:b], y[a:b]), ("masked", Lq, local, e)
    ts.free()
    c.close()
