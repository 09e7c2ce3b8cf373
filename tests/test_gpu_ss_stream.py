"""The ...AndSS kernels (par.ssm = 2 is the reference's default) under everything the stream engine does to a
database (-m gpu): hhv_ss_kernel = workgroups of eight wavefronts sharing one LDS copy of the premultiplied score table, each
wavefront a systolic array drawing segments from the work queue.

tests/test_gpu_ss.py aligns nine templates per case; here the secondary-structure variants meet segment junctions in every
fourth ring chunk, merged short templates, streams shorter than a chunk, fewer segments than the eight arrays of ONE workgroup
(wavefronts that return at once next to wavefronts that work), 1-3-column templates back to back, queries of one, two, three and
more strips (one launch per strip: the carry rows through HBM), the short-query arrays (table values gathered from global
memory), all three table modes, masked rounds (AlignWithCellOffAndSS), local and global - against the oracle, which is pinned
to the reference's -DVITERBI_SS_SCORE builds (tests/test_oracle_vs_reference.py).  Which wavefront of which workgroup aligned a
template must not show."""
import numpy as np
import pytest

from pyoracle import SSInfo, make_params

pytestmark = pytest.mark.gpu

PATTERNS = {
    "L127": ((127,), 3000),                                                               # a junction every fourth chunk
    "mixed": ((63, 63, 500, 31, 31, 31, 31, 1000, 127, 128, 129, 96, 32, 1, 2, 300), 4000),
    "tiny": ((1, 1, 2, 3, 1, 400, 1), 7000),                                              # headers in consecutive steps
    "few": ((300, 64, 200), 5),                                                           # fewer segments than the arrays of one workgroup
    "one_short": ((20,), 1),
}


def ss_inputs(rng, Lq):
    S73 = rng.normal(0, 1, (8, 4, 11)).astype(np.float32)
    S33 = rng.normal(0, 1, (4, 11, 4, 11)).astype(np.float32)
    S37 = rng.normal(0, 1, (4, 11, 8)).astype(np.float32)
    q_ss = (rng.integers(0, 4, Lq + 1), rng.integers(0, 11, Lq + 1), rng.integers(0, 8, Lq + 1))
    return (S73, S33, S37), q_ss


def t_ss_of(rng, L):
    return (rng.integers(0, 4, L + 1), rng.integers(0, 11, L + 1), rng.integers(0, 8, L + 1))


def base_templates(rng, qf, Lq, lengths, seed):
    from pyhhv import synth
    out = {}
    for L in sorted(set(lengths)):
        v = []
        for k in range(3):
            seed += 1
            if k < 2 and L >= 2:
                p, tr = synth.make_homolog(seed, qf, L=L, start=1 + (37 * k + L) % max(1, Lq - min(L, Lq) + 1))
            else:
                p, tr = synth.make_template(seed, L)
            v.append((p, tr, t_ss_of(rng, L)))
        out[L] = v
    return out


@pytest.mark.parametrize("local", [0, 1])
@pytest.mark.parametrize("name", list(PATTERNS))
@pytest.mark.parametrize("Lq,mode", [(300, 4), (300, 2), (161, 1), (431, 4), (700, 4), (150, 4), (70, 2)])
def test_ss_stream(oracle, Lq, mode, name, local):
    from pyhhv import capi, synth
    if (Lq, mode) not in ((300, 4), (431, 4)) and name not in ("mixed", "tiny"):
        pytest.skip("the small patterns run at the headline query length and on a two-strip query")
    cycle, n = PATTERNS[name]
    rng = np.random.default_rng(len(name) * 1000 + Lq * 3 + mode * 7 + local)
    par = make_params(local=local, ss_mode=2)
    qf, qtr = synth.make_query(62000 + Lq, Lq)
    tables, q_ss = ss_inputs(rng, Lq)
    ss = SSInfo(mode, *q_ss, *tables)
    base = base_templates(rng, qf, Lq, cycle, 63000 + Lq)
    Ls = [cycle[k % len(cycle)] for k in range(n)]
    var = rng.integers(0, 3, n)
    tps = [base[L][v][0] for L, v in zip(Ls, var)]
    ttrs = [base[L][v][1] for L, v in zip(Ls, var)]
    t_ss = [base[L][v][2] for L, v in zip(Ls, var)]
    want = {(L, v): oracle.align(par, qf, qtr, base[L][v][0], base[L][v][1], ss=ss, t_ss=base[L][v][2], want_path=True)
            for L in base for v in range(3)}
    w = [want[(L, v)] for L, v in zip(Ls, var)]
    w_score = np.array([a.score for a in w], dtype=np.float32)
    w_i2 = np.array([a.i2 for a in w], dtype=np.int32)
    w_j2 = np.array([a.j2 for a in w], dtype=np.int32)

    c = capi.Context(local=local, ssw=par["ssw"], ss_mode=2)
    c.set_query(qf, qtr)
    c.set_ss_tables(*tables)
    c.set_query_ss(*q_ss)
    c.set_ss_mode(mode)
    ts = c.upload(tps, ttrs, t_ss)

    def check(res, what):
        bad = np.nonzero((res["i2"] != w_i2) | (res["j2"] != w_j2) | (res["score"] != w_score))[0]
        assert bad.size == 0, (what, name, Lq, mode, local, bad.size, bad[:8], res["score"][bad[:4]], w_score[bad[:4]])

    for rep in range(2):
        check(c.align(ts), "score-only %d" % rep)
    res = c.align(ts, backtrace=True)
    check(res, "backtrace")
    hits = c.hits(ts)
    assert np.array_equal(hits["nsteps"], np.array([a.nsteps for a in w], dtype=np.int32))
    assert np.all(hits["score"] == np.array([a.hit_score for a in w], dtype=np.float32))
    assert np.all(hits["score_ss"] == np.array([a.score_ss for a in w], dtype=np.float32))
    sample = sorted(set(list(range(min(n, 12))) + list(range(max(0, n - 12), n)) + [int(e) for e in rng.integers(0, n, 12)]))
    for e in sample:
        a = w[e]
        assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (name, Lq, mode, local, e)
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert ns == a.nsteps and np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(j_s[1:ns + 1], a.j_steps[1:ns + 1])
    # masked round with secondary structure (AlignWithCellOffAndSS)
    masks = {}
    for k, e in enumerate(sorted(set(int(x) for x in rng.integers(0, n, 10)))):
        a = w[e]
        masks[e] = (oracle.exclude_alignment(Lq, Ls[e], a.i_steps, a.j_steps, a.nsteps) if k % 2 == 0
                    else (rng.random((Lq + 1, Ls[e] + 1)) < 0.3).astype(np.uint8))
        c.set_celloff(ts, e, masks[e])
    res2 = c.align(ts, celloff=True)
    for e, m in masks.items():
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=m, ss=ss, t_ss=t_ss[e], want_path=True)
        assert (a.i2, a.j2) == (res2["i2"][e], res2["j2"][e]) and np.float32(a.score) == res2["score"][e], (name, Lq, mode, local, e)
        assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:] & 0x7F, a.bt[1:, 1:] & 0x7F), (name, Lq, mode, local, e)
    others = np.array([e not in masks for e in range(n)])
    assert np.array_equal(res2[others].view(np.uint8), res[others].view(np.uint8))
    # a new query on the same context: its secondary structure replaces the old one's (hhv_set_query forgets it)
    c.set_query(qf, qtr)
    c.set_ss_mode(0)
    c.align(ts)
    c.set_query_ss(*q_ss)
    c.set_ss_mode(mode)
    check(c.align(ts), "after a query change")
    ts.free()
    c.close()


@pytest.mark.parametrize("local", [0, 1])
def test_ss_long_query(oracle, local):
    """Lq 5000: 16 strips, every carry row through HBM, the table shared by eight wavefronts that run different strips' streams"""
    from pyhhv import capi, synth
    Lq, mode = 5000, 4
    rng = np.random.default_rng(77 + local)
    par = make_params(local=local, ss_mode=2)
    qf, qtr = synth.make_query(71000, Lq)
    tables, q_ss = ss_inputs(rng, Lq)
    ss = SSInfo(mode, *q_ss, *tables)
    tps, ttrs, t_ss = [], [], []
    for k, L in enumerate((700, 5000, 130, 1, 2500)):
        p, tr = synth.make_homolog(72000 + k, qf, L=L) if k % 2 == 0 and L >= 2 else synth.make_template(72000 + k, L)
        tps.append(p)
        ttrs.append(tr)
        t_ss.append(t_ss_of(rng, L))
    c = capi.Context(local=local, ssw=par["ssw"], ss_mode=2)
    c.set_query(qf, qtr)
    c.set_ss_tables(*tables)
    c.set_query_ss(*q_ss)
    c.set_ss_mode(mode)
    ts = c.upload(tps, ttrs, t_ss)
    so = c.align(ts)
    res = c.align(ts, backtrace=True)
    hits = c.hits(ts)
    assert np.array_equal(so.view(np.uint8), res.view(np.uint8))
    for e in range(len(tps)):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], ss=ss, t_ss=t_ss[e], want_path=True)
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and np.float32(a.score) == res["score"][e], (local, e)
        assert hits["nsteps"][e] == a.nsteps and np.float32(a.hit_score) == hits["score"][e] and np.float32(a.score_ss) == hits["score_ss"][e]
        if e in (0, 2):
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (local, e)
    ts.free()
    c.close()


@pytest.mark.parametrize("Lq,local", [(431, 0), (431, 1), (512, 1), (640, 0), (700, 1), (1000, 0)])
def test_ss_pairs_equal_one_launch_per_strip(oracle, Lq, local):
    """hhv_ss_pair_kernel (four two-wave pairs per workgroup around one LDS table; chains for more than two strips) against one
    launch per strip (hhv_set_launch_policy pair_mode 0), bit for bit, on sets where every pair walks many segments - junctions,
    the carry FIFO's wrap-around, 1-3-column templates back to back - and against the oracle on the distinct templates"""
    from pyhhv import capi, synth
    rng = np.random.default_rng(4000 + Lq + local)
    par = make_params(local=local, ss_mode=2)
    qf, qtr = synth.make_query(64000 + Lq, Lq)
    tables, q_ss = ss_inputs(rng, Lq)
    ss = SSInfo(4, *q_ss, *tables)
    base = []
    for k in range(40):
        Lt = [1, 1, 2, 3, 1, 300, 64, 129, 130, 511][k % 10] + (k // 10 if k % 10 >= 5 else 0)
        p, tr = synth.make_homolog(65000 + k, qf, L=Lt) if k % 3 == 0 and Lt >= 2 else synth.make_template(65000 + k, Lt)
        base.append((p, tr, t_ss_of(rng, Lt)))
    want = [oracle.align(par, qf, qtr, b[0], b[1], ss=ss, t_ss=b[2], want_path=True) for b in base]
    for n in (20000, 500, 3):
        pick = rng.integers(0, len(base), size=n)
        c = capi.Context(local=local, ssw=par["ssw"], ss_mode=2)
        c.set_query(qf, qtr)
        c.set_ss_tables(*tables)
        c.set_query_ss(*q_ss)
        c.set_ss_mode(4)
        ts = c.upload([base[p][0] for p in pick], [base[p][1] for p in pick], [base[p][2] for p in pick])
        got = []
        for mode in (1, 0):
            c.set_launch_policy(pair_mode=mode)
            so = c.align(ts).copy()
            res = c.align(ts, backtrace=True).copy()
            hits = c.hits(ts).copy()
            got.append((so, res, hits))
        c.set_launch_policy(pair_mode=-1)
        (so1, res1, hits1), (so0, res0, hits0) = got
        assert so1.tobytes() == so0.tobytes() and res1.tobytes() == res0.tobytes() and hits1.tobytes() == hits0.tobytes(), (Lq, local, n)
        for e in range(min(n, 400)):
            a = want[pick[e]]
            assert (a.i2, a.j2) == (res1["i2"][e], res1["j2"][e]) and np.float32(a.score) == res1["score"][e] == so1["score"][e], (Lq, local, n, e)
            assert hits1["nsteps"][e] == a.nsteps and np.float32(a.hit_score) == hits1["score"][e] and np.float32(a.score_ss) == hits1["score_ss"][e]
        ts.free()
        c.close()
