"""Deterministic adversarial parity on the 64-lane systolic arrays (-m gpu).

What the seeded workloads of test_gpu_parity.py reach only by chance: headers of 1-3-column templates BACK TO BACK inside one
wavefront's stream range (the finalized best of a template travels lane to lane through 8-byte LDS slots and is read "one
step later", hhv_stream_kernel.h header_tid_best / publish_best: with a header in almost every step every lane has a hand-off
in flight all the time), exact ties between DP candidates (transition scores on a coarse grid: the strict '>' tie-break of
src/hhviterbialgorithm.cpp:423-455,462-486 and the order of the five MM candidates decide), -100000 transitions in the middle
of a profile, the column-0 reset right behind a template's last column (:161-173).

Lq = 161 (R = 3 rows per lane), 300 (R = 5, the headline kernel), 431 (two passes, 4 + 3 rows: carry rows and the running best
cross the pass boundary for every one of the tiny templates); ~21 000 templates per case so that each of the 2048 resident
waves walks ten of them back to back; lengths cycle 1, 1, 2, 3, 1, long, 1; local and global; score-only, backtrace (every
byte of a sample, paths, Hit scores of all) and a masked round.  Everything against the oracle; the distinct templates also
against the reference's own Viterbi::Align (oracle/_ref) where it is built."""
import numpy as np
import pytest

from pyoracle import Ref, have_ref, make_params

pytestmark = pytest.mark.gpu

CYCLE = (1, 1, 2, 3, 1, 0, 1)          # 0 = a long template
LONG = (300, 257, 64, 130, 301, 33)
N_TEMPLATES = 21000


@pytest.fixture(scope="module")
def hhv():
    from pyhhv import capi
    capi.load()
    return capi


def grid(a, step=0.5):
    """transition scores on a coarse grid -> exact ties between candidates; impossible transitions stay impossible"""
    g = (np.round(a / step) * step).astype(np.float32)
    g[a < -1000] = -100000.0
    return g


def make_base(rng, qf, Lq):
    """6 variants of every slot of the length cycle: tiny templates (random and cut out of the query), long ones mostly homologous"""
    from pyhhv import synth
    base = [[] for _ in CYCLE]
    seed = 40000 + Lq
    for slot, L in enumerate(CYCLE):
        for v in range(6):
            Lt = L if L else LONG[v]
            seed += 1
            homolog = (v % 2 == 0) if L else (v != 3)
            p, tr = synth.make_homolog(seed, qf, L=Lt, start=1 + (7 * v) % max(1, Lq - Lt)) if homolog else synth.make_template(seed, Lt)
            tr = grid(tr)
            if v % 3 == 1 and Lt >= 2:      # a dead transition in the middle of the profile
                tr[int(rng.integers(1, Lt)), int(rng.integers(0, 7))] = -100000.0
            if v == 5 and Lt >= 3:          # ... and a column nothing can leave as a match
                tr[Lt // 2, 0] = -100000.0
            base[slot].append((p, tr))
    return base


@pytest.mark.parametrize("local", [0, 1])
@pytest.mark.parametrize("Lq", [161, 300, 431])
def test_back_to_back_tiny_templates_ties_dead_transitions(hhv, oracle, Lq, local):
    from pyhhv import synth
    rng = np.random.default_rng(Lq * 2 + local)
    par = make_params(local=local, egq=0.0 if Lq != 161 else 0.2, egt=0.0 if Lq != 161 else 0.1)
    qf, qtr = synth.make_query(31000 + Lq, Lq)
    qtr = grid(qtr)
    qtr[Lq // 3, 2] = -100000.0          # no M->D out of one query row, no I->I in another
    qtr[2 * Lq // 3, 4] = -100000.0
    base = make_base(rng, qf, Lq)
    n = N_TEMPLATES
    slot = np.arange(n) % len(CYCLE)
    var = rng.integers(0, 6, n)
    tps = [base[s][v][0] for s, v in zip(slot, var)]
    ttrs = [base[s][v][1] for s, v in zip(slot, var)]
    want = {(s, v): oracle.align(par, qf, qtr, base[s][v][0], base[s][v][1], want_path=True) for s in range(len(CYCLE)) for v in range(6)}
    if have_ref():
        # the distinct templates against the reference's own Viterbi::Align, one per SIMD batch (MapOneHMM: the parity definition)
        ref = Ref()
        for (s, v), a in want.items():
            r = ref.align_batch(par, qf, qtr, [base[s][v][0]], [base[s][v][1]], replicate=True, want_path=True)[0]
            assert (r.i2, r.j2) == (a.i2, a.j2) and np.float32(r.score).tobytes() == np.float32(a.score).tobytes(), (s, v)
            assert np.array_equal(r.bt[1:, 1:] & 0x7F, a.bt[1:, 1:] & 0x7F) and r.nsteps == a.nsteps, (s, v)
            assert np.float32(r.hit_score).tobytes() == np.float32(a.hit_score).tobytes(), (s, v)

    c = hhv.Context(local=par["local"], egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"],
                    ssw=par["ssw"], ss_mode=par["ss_mode"])
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    w_score = np.array([want[(s, v)].score for s, v in zip(slot, var)], dtype=np.float32)
    w_i2 = np.array([want[(s, v)].i2 for s, v in zip(slot, var)], dtype=np.int32)
    w_j2 = np.array([want[(s, v)].j2 for s, v in zip(slot, var)], dtype=np.int32)

    def check(res, what):
        bad = np.nonzero((res["i2"] != w_i2) | (res["j2"] != w_j2) | (res["score"].view(np.uint32) != w_score.view(np.uint32)))[0]
        # (+0 / -0 are the same score: v_max_f32 vs MAXPS, DESIGN.md 3 parity notes)
        bad = [e for e in bad if not (res["i2"][e] == w_i2[e] and res["j2"][e] == w_j2[e] and res["score"][e] == w_score[e])]
        assert not bad, (what, Lq, local, len(bad), bad[:8], [(res[e], w_score[e], w_i2[e], w_j2[e]) for e in bad[:3]])

    plain = c.align(ts)
    check(plain, "score-only")
    res = c.align(ts, backtrace=True)
    check(res, "backtrace")
    hits = c.hits(ts)
    w_ns = np.array([want[(s, v)].nsteps for s, v in zip(slot, var)], dtype=np.int32)
    w_hit = np.array([want[(s, v)].hit_score for s, v in zip(slot, var)], dtype=np.float32)
    assert np.array_equal(hits["nsteps"], w_ns)
    assert np.all(hits["score"] == w_hit)
    # every backtrace byte and the path of a sample: the neighbours of long templates, a block in the middle, both ends
    sample = sorted(set(list(range(0, 16)) + list(range(n // 2, n // 2 + 16)) + list(range(n - 16, n)) +
                        [int(e) for e in rng.integers(0, n, 24)]))
    for e in sample:
        a = want[(slot[e], var[e])]
        assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (Lq, local, e)
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert ns == a.nsteps and np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(j_s[1:ns + 1], a.j_steps[1:ns + 1])
        assert np.array_equal(st[1:ns + 1], a.states[1:ns + 1]) and np.array_equal(S[1:ns + 1], a.S[1:ns + 1])
    # masked round: the first alignment of some templates switched off (tiny ones and long ones, neighbours of each other),
    # a random mask for a few more; every other template must come out as before
    masked = {}
    for e in [5, 6, 7, 8, n // 2 + 3, n // 2 + 4, n // 2 + 5, n - 9, n - 8]:
        a = want[(slot[e], var[e])]
        masked[e] = oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, a.i_steps, a.j_steps, a.nsteps)
    for e in [20, 21, 26, n - 30]:
        m = (rng.random((Lq + 1, tps[e].shape[0])) < 0.3).astype(np.uint8)
        masked[e] = m
    for e, m in masked.items():
        c.set_celloff(ts, e, m)
    res2 = c.align(ts, celloff=True)
    for e in range(n):
        if e in masked:
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=masked[e], want_path=True)
            assert (a.i2, a.j2) == (res2["i2"][e], res2["j2"][e]) and np.float32(a.score) == res2["score"][e], (Lq, local, e)
            assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:] & 0x7F, a.bt[1:, 1:] & 0x7F), (Lq, local, e)
    others = np.array([e not in masked for e in range(n)])
    assert np.array_equal(res2[others].view(np.uint8), res[others].view(np.uint8))
    # determinism
    again = c.align(ts, backtrace=True)
    assert np.array_equal(again.view(np.uint8), res.view(np.uint8))
    ts.free()
    c.close()


@pytest.mark.parametrize("local", [0, 1])
@pytest.mark.parametrize("Lq", [80, 300, 431])
def test_subnormal_emission_products(hhv, oracle, Lq, local):
    """Profile values around 2^-70: every product q.p[i][a] * t.p[j][a] of ScalarProd20Vec (src/hhviterbialgorithm.cpp:263-292
    through simd.h) lies in [2^-149, 2^-126], i.e. is a SUBNORMAL float, and so is their sum - a build that flushed fp32
    denormals (the kernels are compiled with .amdhsa_float_denorm_mode_32 3 = keep) would feed log2f4 a zero instead
    (tools/soak.py's smallest values, 2^-99.999, have products that are zero in every arithmetic).  Score-only,
    backtrace bytes, paths and Hit scores against the oracle and - where built - the reference's own Viterbi::Align."""
    from pyhhv import synth
    rng = np.random.default_rng(9000 + Lq + local)
    par = make_params(local=local, shift=127.8)   # log2 of a column sum is ~ -128 +- 3: with this offset the alignments are real ones
    qf, qtr = synth.make_query(52000 + Lq, Lq)
    qf = (qf * np.float32(2.0 ** -66)).astype(np.float32)       # entries ~2^-70 .. 2^-66
    lens = [1, 2, 3, 17, 64, 65, 130, 257, 300, 330]
    tps, ttrs = [], []
    for k, Lt in enumerate(lens):
        p, tr = synth.make_homolog(53000 + 7 * k + Lq, (qf * np.float32(2.0 ** 66)).astype(np.float32), L=Lt) if k % 2 == 0 else synth.make_template(53000 + 7 * k + Lq, Lt)
        scale = np.float32(2.0 ** -float(rng.integers(68, 72)))
        tps.append((p * scale).astype(np.float32))
        ttrs.append(tr)
    # every product is subnormal and none is zero
    assert np.float64(qf[1:].max()) * np.float64(max(t[1:].max() for t in tps)) < 2.0 ** -126
    assert np.float64(qf[1:].min()) * np.float64(min(t[1:].min() for t in tps)) > 2.0 ** -149
    c = hhv.Context(local=par["local"], egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"],
                    ssw=par["ssw"], ss_mode=par["ss_mode"])
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    res0 = c.align(ts)
    res = c.align(ts, backtrace=True)
    hits = c.hits(ts)
    ref = Ref() if have_ref() else None
    for e in range(len(lens)):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_path=True)
        for r_, what in ((res0, "score-only"), (res, "backtrace")):
            assert (a.i2, a.j2) == (r_["i2"][e], r_["j2"][e]), (what, Lq, local, e)
            assert np.float32(a.score) == r_["score"][e], (what, Lq, local, e, a.score, r_["score"][e])
        assert np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:], a.bt[1:, 1:]), (Lq, local, e)
        assert hits["nsteps"][e] == a.nsteps and hits["score"][e] == np.float32(a.hit_score)
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(S[1:ns + 1], a.S[1:ns + 1])
        if ref is not None:
            r = ref.align_batch(par, qf, qtr, [tps[e]], [ttrs[e]], replicate=True, want_path=True)[0]
            assert (r.i2, r.j2) == (a.i2, a.j2) and np.float32(r.score).tobytes() == np.float32(a.score).tobytes(), (Lq, local, e)
            assert np.array_equal(r.bt[1:, 1:] & 0x7F, a.bt[1:, 1:] & 0x7F), (Lq, local, e)
    ts.free()
    c.close()
