"""GPU parity (run with -m gpu on the MI355X box): the HIP engine, called through the C ABI, against
the oracle on the same seeded inputs.  Integer outputs (endpoints, backtrace bytes, paths) are
bit-exact; float scores are compared as IEEE values and additionally within the north-star
tolerance of 1e-4."""
import numpy as np
import pytest

from common import same_float, workload
from pyoracle import make_params

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: scores within 1e-4 of the reference CPU path


@pytest.fixture(scope="module")
def hhv():
    from pyhhv import capi
    capi.load()
    return capi


def ctx_for(hhv, par):
    return hhv.Context(local=par["local"], egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"],
                       ssw=par["ssw"], ss_mode=par["ss_mode"])


@pytest.mark.parametrize("case", range(16))
def test_score_only_matches_oracle(hhv, oracle, case):
    rng = np.random.default_rng(case)
    Lq = [5, 64, 65, 128, 200, 256, 257, 300, 320, 321, 431, 512, 700, 961, 1300, 2000][case]
    par = make_params(local=case % 2, egq=0.0 if case % 4 < 2 else 0.3, egt=0.0 if case % 4 < 2 else 0.1)
    n = int(rng.integers(3, 40))
    qf, qtr, tps, ttrs = workload(case, Lq, n, 1, 260)
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    res = c.align(ts)
    for e in range(n):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=False)
        assert res["index"][e] == e
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]), (case, e)
        assert same_float(a.score, res["score"][e]) and abs(float(a.score) - float(res["score"][e])) <= TOL
    ts.free()
    c.close()


@pytest.mark.parametrize("case", range(8))
def test_backtrace_hits_match_oracle(hhv, oracle, case):
    rng = np.random.default_rng(50 + case)
    Lq = [40, 100, 200, 300, 320, 431, 650, 1100][case]
    par = make_params(local=case % 2, egq=0.0 if case % 4 < 2 else 0.2, egt=0.0 if case % 4 < 2 else 0.1)
    n = int(rng.integers(4, 24))
    qf, qtr, tps, ttrs = workload(50 + case, Lq, n, 5, 330)
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    res = c.align(ts, backtrace=True)
    # hhv_backtrace (Viterbi::Backtrace of one template) straight after the alignment: it starts the device walk itself
    a0 = oracle.align(par, qf, qtr, tps[0], ttrs[0], want_path=True)
    ns0, mc0, bi, bj, bs = c.backtrace(ts, 0)
    assert (ns0, mc0) == (a0.nsteps, a0.matched_cols)
    assert np.array_equal(bi[1:ns0 + 1], a0.i_steps[1:ns0 + 1]) and np.array_equal(bj[1:ns0 + 1], a0.j_steps[1:ns0 + 1])
    assert np.array_equal(bs[1:ns0 + 1], a0.states[1:ns0 + 1])
    hits = c.hits(ts)
    for e in range(n):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_path=True)
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e])
        m = c.backtrace_matrix(ts, e)
        assert np.array_equal(m[1:, 1:], a.bt[1:, 1:]), (case, e)
        h = hits[e]
        assert h["index"] == e and h["nsteps"] == a.nsteps and h["matched_cols"] == a.matched_cols
        assert (h["i1"], h["j1"]) == (a.i_steps[a.nsteps], a.j_steps[a.nsteps])
        assert same_float(h["score"], a.hit_score) and abs(float(h["score"]) - float(a.hit_score)) <= TOL
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert ns == a.nsteps
        assert np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1])
        assert np.array_equal(j_s[1:ns + 1], a.j_steps[1:ns + 1])
        assert np.array_equal(st[1:ns + 1], a.states[1:ns + 1])
        assert np.array_equal(S[1:ns + 1], a.S[1:ns + 1])
    # top-K: by hit score descending, ties by index
    k = min(5, n)
    top, nv = c.topk(ts, k)
    order = sorted(range(n), key=lambda e: (-float(hits["score"][e]), e))[:k]
    assert nv == k and list(top["index"]) == order
    ts.free()
    c.close()


def test_celloff_second_round(hhv, oracle):
    """Alt-alignment round 2 (src/hhviterbirunner.cpp:104,152-164): mask the first path, re-align."""
    for local, Lq in ((0, 150), (1, 150), (1, 400), (0, 700)):
        par = make_params(local=local)
        qf, qtr, tps, ttrs = workload(40 + local, Lq, 5, 80, 170, homolog_every=1)
        c = ctx_for(hhv, par)
        c.set_query(qf, qtr)
        ts = c.upload(tps, ttrs)
        c.align(ts, backtrace=True)
        c.hits(ts)
        masks = []
        for e in range(5):
            ns, i_s, j_s, st, S = c.hit_path(ts, e)
            m = oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, i_s, j_s, ns)
            masks.append(m)
        for e in range(5):
            c.set_celloff(ts, e, masks[e])
        res = c.align(ts, celloff=True)
        hits = c.hits(ts)
        for e in range(5):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=masks[e], want_path=True)
            assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e])
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:])
            assert same_float(hits["score"][e], a.hit_score) and hits["nsteps"][e] == a.nsteps
        ts.free()
        c.close()


def test_celloff_for_a_subset_after_a_backtrace_run(hhv, oracle):
    """hhv_set_celloff for SOME templates after a backtrace launch: the mask bytes share the buffer with the compare bits
    the launch left behind, so the templates without a fresh mask must read as "no cell excluded", not as stale bits."""
    for local, Lq in ((1, 150), (0, 431)):
        par = make_params(local=local)
        qf, qtr, tps, ttrs = workload(90 + local, Lq, 6, 80, 170, homolog_every=1)
        c = ctx_for(hhv, par)
        c.set_query(qf, qtr)
        ts = c.upload(tps, ttrs)
        plain = c.align(ts, backtrace=True)
        c.hits(ts)
        masked = {}
        for e in (0, 3):
            ns, i_s, j_s, st, S = c.hit_path(ts, e)
            masked[e] = oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, i_s, j_s, ns)
        for e, m in masked.items():
            c.set_celloff(ts, e, m)
        res = c.align(ts, celloff=True)
        for e in range(6):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=masked.get(e), want_path=True)
            assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e]), (local, e)
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:])
            if e not in masked:
                assert res[e] == plain[e]
        # and again: the second masked launch left compare bits too; "NULL clears" for one template, the rest unmasked
        c.set_celloff(ts, 3, None)
        res = c.align(ts, celloff=True)
        assert np.array_equal(res.view(np.uint8), plain.view(np.uint8))
        ts.free()
        c.close()


def test_strip_plan_rows_per_pass(hhv, oracle):
    """Queries whose passes have different rows per lane (Lq 431: 4 + 3, 385: 4 + 3, 700: 4 + 4 + 3, 1000: 4 x 4,
    1217: 4 x 5): scores, backtrace bytes of every pass and paths against the oracle."""
    for Lq in (385, 431, 700, 1217):
        par = make_params(local=Lq % 2)
        qf, qtr, tps, ttrs = workload(300 + Lq, Lq, 4, 60, 200, homolog_every=1)
        c = ctx_for(hhv, par)
        c.set_query(qf, qtr)
        ts = c.upload(tps, ttrs)
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        for e in range(4):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_path=True)
            assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e]), (Lq, e)
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (Lq, e)
            ns, i_s, j_s, st, S = c.hit_path(ts, e)
            assert ns == a.nsteps and np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(j_s[1:ns + 1], a.j_steps[1:ns + 1])
            assert same_float(hits["score"][e], a.hit_score)
        ts.free()
        c.close()


@pytest.mark.parametrize("Lq", [1, 7, 16, 17, 33, 64, 80, 81, 97, 128, 160, 161])
def test_short_query_arrays(hhv, oracle, Lq):
    """Short queries run as 4 (Lq <= 80) or 2 (Lq <= 160) independent systolic arrays per wavefront, every array on its own
    stream range: more templates than arrays so that every array of every wave works, ragged lengths (templates shorter
    than an array, streams longer than the ring), with backtrace, Hit scores and paths; a masked second round for some."""
    rng = np.random.default_rng(Lq)
    par = make_params(local=Lq % 2, egq=0.0 if Lq % 3 else 0.2, egt=0.0 if Lq % 3 else 0.1)
    n = 700
    from pyhhv import synth
    qf, qtr = synth.make_query(3000 + Lq, Lq)
    base = []
    for k in range(48):
        L = int(rng.integers(1, 200)) if k % 6 else int(rng.integers(1, 12))
        base.append(synth.make_homolog(9100 + k, qf, L=L) if k % 3 == 0 else synth.make_template(9100 + k, L))
    idx = rng.integers(0, 48, n)
    tps = [base[i][0] for i in idx]
    ttrs = [base[i][1] for i in idx]
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    plain = c.align(ts)
    res = c.align(ts, backtrace=True)
    assert np.array_equal(plain.view(np.uint8), res.view(np.uint8))
    hits = c.hits(ts)
    ref = [oracle.align(par, qf, qtr, base[i][0], base[i][1], want_path=True) for i in range(48)]
    for e in range(n):
        a = ref[idx[e]]
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e]), (Lq, e)
        assert hits["nsteps"][e] == a.nsteps and same_float(hits["score"][e], a.hit_score), (Lq, e)
    for e in (0, 1, n // 2, n - 1):
        a = ref[idx[e]]
        assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (Lq, e)
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(S[1:ns + 1], a.S[1:ns + 1])
    # cell-off round for three templates, the others unmasked
    masked = {}
    for e in (2, n // 3, n - 2):
        a = ref[idx[e]]
        masked[e] = oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, a.i_steps, a.j_steps, a.nsteps)
        c.set_celloff(ts, e, masked[e])
    res2 = c.align(ts, celloff=True)
    for e in range(n):
        if e in masked:
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=masked[e], want_path=True)
            assert (a.i2, a.j2) == (res2["i2"][e], res2["j2"][e]) and same_float(a.score, res2["score"][e]), (Lq, e)
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:])
        else:
            assert res2[e] == res[e]
    ts.free()
    c.close()


def test_many_templates_partitioning(hhv, oracle):
    """More templates than resident waves, ragged lengths: exercises the wave partition, chunk refills
    (streams much longer than the 192-record LDS ring) and the header/finalize plumbing."""
    par = make_params(local=0)
    Lq = 300
    n = 3000
    rng = np.random.default_rng(7)
    from pyhhv import synth
    qf, qtr = synth.make_query(4242, Lq)
    base = [synth.make_template(9000 + k, int(rng.integers(20, 400))) for k in range(64)]
    idx = rng.integers(0, 64, n)
    tps = [base[i][0] for i in idx]
    ttrs = [base[i][1] for i in idx]
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    res = c.align(ts)
    ref = [oracle.align(par, qf, qtr, base[i][0], base[i][1], want_bt=False) for i in range(64)]
    for e in range(n):
        a = ref[idx[e]]
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e]), e
    # determinism: a second run is bitwise identical
    res2 = c.align(ts)
    assert np.array_equal(res.view(np.uint8), res2.view(np.uint8))
    ts.free()
    c.close()


def test_headline_shape_sample(hhv, oracle):
    """BASELINE configs[1] shape (Lq=300 vs Lt=300, global, score-only): a seeded sample of templates
    checked against the oracle, plus size-independent properties on the whole set."""
    par = make_params(local=0)
    from pyhhv import synth
    Lq = Lt = 300
    n = 2048
    qf, qtr = synth.make_query(1, Lq)
    tps, ttrs = [], []
    for k in range(n):
        kk = k % 256
        p, tr = synth.make_template(100 + kk, Lt) if kk % 3 else synth.make_homolog(100 + kk, qf, L=Lt)
        tps.append(p)
        ttrs.append(tr)
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    res = c.align(ts)
    for e in list(range(0, n, 97)):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=False)
        assert (a.i2, a.j2) == (res["i2"][e], res["j2"][e]) and same_float(a.score, res["score"][e])
    # identical templates -> identical results wherever they sit in the stream
    for f in ("score", "i2", "j2"):
        assert np.array_equal(res[f][256:], res[f][:-256])
    # global alignment ends in the last row or the last column
    assert np.all((res["i2"] == Lq) | (res["j2"] == Lt))
    ts.free()
    c.close()


def test_packed_db_roundtrip(hhv, oracle, tmp_path):
    """hhv_db_write -> hhv_db_open gives the same results as uploading the profiles directly."""
    par = make_params(local=1)
    qf, qtr, tps, ttrs = workload(91, 120, 12, 30, 200)
    path = str(tmp_path / "db.hhvpdb")
    Ls = hhv.db_write(path, tps, ttrs)
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    a = c.upload(tps, ttrs)
    b = c.db_open(path, Ls)
    ra, rb = c.align(a), c.align(b)
    assert np.array_equal(ra.view(np.uint8), rb.view(np.uint8))
    o = oracle.align(par, qf, qtr, tps[3], ttrs[3], want_bt=False)
    assert (o.i2, o.j2) == (rb["i2"][3], rb["j2"][3]) and same_float(o.score, rb["score"][3])
    a.free()
    b.free()
    c.close()


def test_gather_subset_on_device(hhv, oracle):
    """hhv_tset_gather: a device-side copy of selected templates aligns exactly like the originals."""
    par = make_params(local=1)
    qf, qtr, tps, ttrs = workload(77, 150, 12, 5, 260)
    c = ctx_for(hhv, par)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    full = c.align(ts, backtrace=True)
    ids = np.array([11, 0, 5, 5, 3], np.int32)
    sub = c.gather(ts, ids)
    res = c.align(sub, backtrace=True)
    assert list(res["index"]) == list(range(len(ids)))
    assert np.array_equal(res["score"].view(np.uint32), full["score"][ids].view(np.uint32))
    assert np.array_equal(res["i2"], full["i2"][ids]) and np.array_equal(res["j2"], full["j2"][ids])
    for pos, k in enumerate(ids):
        assert np.array_equal(c.backtrace_matrix(sub, pos), c.backtrace_matrix(ts, int(k)))
    sub.free()
    ts.free()
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("Lq", [37, 150, 400])
def test_global_mode_simd_batch_last_column(oracle, Lq):
    """hhv_set_global_batch: the reference maximises the global score over the last column of the LONGEST template of a SIMD
    batch (src/hhviterbialgorithm.cpp:462-486); for the shorter templates of the batch only the last row counts.  Batches
    of 8 consecutive templates of the length-sorted list, as ViterbiRunner builds them (src/hhviterbirunner.cpp:117-151);
    the oracle emulates the batch with Lbatch (pinned to the reference, tests/test_oracle_vs_reference.py).  Multi-pass
    queries (Lq = 400) and the backtrace included; clearing the marks restores the single-template result."""
    from pyhhv import capi
    par = make_params(local=0)
    qf, qtr, tps, ttrs = workload(300 + Lq, Lq, 29, 20, 260, homolog_every=2)
    order = sorted(range(len(tps)), key=lambda k: -(tps[k].shape[0] - 1))
    tps, ttrs = [tps[k] for k in order], [ttrs[k] for k in order]
    Ls = [t.shape[0] - 1 for t in tps]
    Lb = [max(Ls[b - b % 8:b - b % 8 + 8]) for b in range(len(Ls))]
    c = capi.Context(local=0)
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    alone = c.align(ts, backtrace=True).copy()
    c.set_global_batch(ts, np.asarray([L < b for L, b in zip(Ls, Lb)], dtype=np.uint8))
    res = c.align(ts, backtrace=True)
    hits = c.hits(ts)
    changed = 0
    for k in range(len(tps)):
        a = oracle.align(par, qf, qtr, tps[k], ttrs[k], Lbatch=Lb[k], want_path=True)
        assert (a.i2, a.j2) == (int(res["i2"][k]), int(res["j2"][k])), (k, Ls[k], Lb[k], a.i2, a.j2, res[k])
        assert same_float(a.score, res["score"][k]), (k, a.score, res["score"][k])
        assert same_float(a.hit_score, hits["score"][k]) and a.nsteps == hits["nsteps"][k], k
        changed += (int(res["i2"][k]), int(res["j2"][k])) != (int(alone["i2"][k]), int(alone["j2"][k]))
    assert changed > 0, "the workload must contain a template whose own last column wins when aligned alone"
    c.set_global_batch(ts, None)
    again = c.align(ts, backtrace=True)
    assert np.array_equal(again, alone)
    ts.free()
    c.close()
