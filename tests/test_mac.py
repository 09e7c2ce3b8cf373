"""MAC realignment: PosteriorDecoder::realign = forward / backward / MAC DP / MAC backtrace
(/root/reference src/hhposteriordecoder.cpp:86-119 and the four algorithm files).
CPU: the oracle restatement against the reference's own member functions (oracle/ref_mac_harness.cpp), bit for bit.
GPU: the HIP kernels against the oracle."""
import numpy as np
import pytest

from pyhhv import synth
from pyoracle import (mac_list_profile, mac_plane_to_list, mac_posterior_list, make_params, oracle_mac_realign,
                      ref_mac_realign)

CASES = [
    # (Lq, Lt, local, homolog?, mact)
    (120, 100, 1, True, 0.3501),
    (120, 100, 0, True, 0.3501),
    (63, 200, 1, True, 0.1),
    (200, 64, 0, True, 0.5),
    (150, 150, 1, False, 0.3501),
    (33, 31, 1, True, 0.0),
    (300, 320, 1, True, 0.3501),
]


def make_pair(case, seed=0):
    Lq, Lt, local, hom, mact = CASES[case]
    qp, qtr = synth.make_query(1000 + 7 * case + seed, Lq)
    if hom:
        tp, ttr = synth.make_homolog(2000 + 7 * case + seed, qp, L=Lt)
    else:
        tp, ttr = synth.make_template(2000 + 7 * case + seed, Lt)
    return qp, qtr, tp, ttr, local, mact


def lin_query(qtr):
    """Log2LinTransitionProbs(1.0) (powf, src/hhhmm.cpp:2305-2313) + initializeQueryHMMTransitions
    (src/hhposteriordecoderrunner.cpp:146-155); test helper - in the product the caller's HMM code does this."""
    t = np.exp2(qtr.astype(np.float32)).astype(np.float32)
    t[0, [synth.M2D, synth.M2I, synth.I2M, synth.I2I, synth.D2M, synth.D2D]] = 0.0
    t[-1] = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    return t


def lin_template(ttr):
    """Log2LinTransitionProbs(1.0) + the boundary assignments of initializeForAlignment (src/hhposteriordecoder.cpp:159-167)."""
    t = np.exp2(ttr.astype(np.float32)).astype(np.float32)
    t[0] = [1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    t[-1] = [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]
    return t


def oracle_mac_realign_nomask(orc, qp, q_lin, tp, t_lin, local=1, mact=0.3501):
    """The four stages on an all-on mask (what the product does for celloff = NULL)."""
    class V:
        pass
    v = V()
    Lq, Lt = qp.shape[0] - 1, tp.shape[0] - 1
    # a fake Viterbi hit whose band covers everything: i1 = j1 = huge, i2 = j2 = 0 -> every cell is on
    v.nsteps, v.i2, v.j2 = 0, 0, 0
    v.i_steps = np.array([Lq + Lt + 5], np.int32)
    v.j_steps = np.array([Lq + Lt + 5], np.int32)
    return oracle_mac_realign(orc, qp, q_lin, tp, t_lin, v, local=local, mact=mact)


def same(a, b):
    assert a.nsteps == b.nsteps and (a.i1, a.j1, a.i2, a.j2, a.matched_cols) == (b.i1, b.j1, b.i2, b.j2, b.matched_cols)
    n = a.nsteps
    lo = 0 if n == 0 else 1   # entry 0 is only written when the backtrace does not start in a match state
    assert np.array_equal(a.i_steps[lo:n + 1], b.i_steps[lo:n + 1]) and np.array_equal(a.j_steps[lo:n + 1], b.j_steps[lo:n + 1])
    assert np.array_equal(a.states[1:n + 1], b.states[1:n + 1])
    assert a.S[1:n + 1].tobytes() == b.S[1:n + 1].tobytes() and a.P[1:n + 1].tobytes() == b.P[1:n + 1].tobytes()
    assert np.float32(a.sum_of_probs).tobytes() == np.float32(b.sum_of_probs).tobytes()


def lists_of(o):
    """the three lists of an oracle result (dense planes -> entries in the reference's order)"""
    return [mac_plane_to_list(o.fwd_list), mac_plane_to_list(o.bwd_list), mac_posterior_list(o.posterior, o.celloff, o.i_steps, o.j_steps, o.nsteps)]


def same_lists(got, r, Lq):
    """got: three (i, j, value) lists; r: a reference result with .lists / .profiles"""
    for w in range(3):
        assert np.array_equal(got[w][0], r.lists[w][0]) and np.array_equal(got[w][1], r.lists[w][1]), (w, len(got[w][0]), len(r.lists[w][0]))
        assert got[w][2].tobytes() == r.lists[w][2].tobytes(), w
    for w in range(2):
        assert mac_list_profile(got[w], Lq).tobytes() == r.profiles[w].tobytes(), w


@pytest.mark.parametrize("case", range(len(CASES)))
def test_oracle_mac_matches_reference(oracle, ref, case):
    qp, qtr, tp, ttr, local, mact = make_pair(case)
    par = make_params(local=local, ss_mode=0)
    vit = oracle.align(par, qp, qtr, tp, ttr, want_path=True)
    prev = []
    for rnd in range(3):   # the first alignment and two alternative ones (alt_i/alt_j of the earlier rounds excluded)
        r = ref_mac_realign(ref, qp, qtr, tp, ttr, vit, local=local, mact=mact, prev=prev)
        o = oracle_mac_realign(oracle, qp, r.q_tr_lin, tp, r.t_tr_lin, vit, local=local, mact=mact, prev=prev)
        assert np.array_equal(o.celloff, r.celloff)
        assert o.scale.tobytes() == r.scale.tobytes()
        assert o.forward[1:].tobytes() == r.forward[1:].tobytes()
        assert np.float64(o.Pforward).tobytes() == np.float64(r.Pforward).tobytes()
        assert o.posterior[1:, 1:].tobytes() == r.posterior[1:, 1:].tobytes()
        assert np.array_equal(o.bmm[1:, 1:], r.bmm[1:, 1:])
        same(o, r)
        # the -o_matrices lists (writeProfilesToHits): forward, backward, posterior entries and the two profiles
        same_lists(lists_of(o), r, o.forward.shape[0] - 1)
        prev.append((r.i_steps[(0 if r.nsteps == 0 else 1):r.nsteps + 1].copy(), r.j_steps[(0 if r.nsteps == 0 else 1):r.nsteps + 1].copy()))
    assert r.nsteps >= 0


@pytest.mark.parametrize("case", range(len(CASES)))
def test_host_mask_and_linear_transitions_match_reference(oracle, ref, case):
    """hhv::LinearTransitions and hhv::MacCellOff (host layer of the product) against the reference's realign()."""
    from pyhhv import capi
    qp, qtr, tp, ttr, local, mact = make_pair(case)
    vit = oracle.align(make_params(local=local, ss_mode=0), qp, qtr, tp, ttr, want_path=True)
    Lq, Lt = qp.shape[0] - 1, tp.shape[0] - 1
    prev = []
    for rnd in range(3):
        r = ref_mac_realign(ref, qp, qtr, tp, ttr, vit, local=local, mact=mact, prev=prev)
        assert capi.linear_transitions(qtr, True).tobytes() == r.q_tr_lin.tobytes()
        assert capi.linear_transitions(ttr, False).tobytes() == r.t_tr_lin.tobytes()
        ns = vit.nsteps
        hit = (int(vit.i_steps[ns]), int(vit.j_steps[ns]), vit.i2, vit.j2, ns, vit.i_steps, vit.j_steps)
        assert np.array_equal(capi.mac_celloff(Lq, Lt, hit, prev), r.celloff)
        lo = 0 if r.nsteps == 0 else 1
        prev.append((r.i_steps[lo:r.nsteps + 1].copy(), r.j_steps[lo:r.nsteps + 1].copy()))


@pytest.mark.gpu
@pytest.mark.parametrize("local", [1, 0])
def test_gpu_runner_matches_reference(oracle, ref, local):
    """hhv::PosteriorDecoderRunner::executeComputation (masks + rounds on the host, DP on the GPU) against the reference's
    realign() chain: five templates, three alternative alignments each, handed over in shuffled order."""
    from pyhhv import capi
    Lq = 140
    qp, qtr = synth.make_query(4242, Lq)
    q_lin = capi.linear_transitions(qtr, True)
    tps, t_lins, hits, want = [], [], [], []
    for k, Lt in enumerate([90, 150, 64, 201, 33]):
        tp, ttr = synth.make_homolog(900 + k, qp, L=Lt)
        vit = oracle.align(make_params(local=local, ss_mode=0), qp, qtr, tp, ttr, want_path=True)
        ns = vit.nsteps
        prev = []
        for irep in (1, 2, 3):
            r = ref_mac_realign(ref, qp, qtr, tp, ttr, vit, local=local, prev=prev)
            lo = 0 if r.nsteps == 0 else 1
            prev.append((r.i_steps[lo:r.nsteps + 1].copy(), r.j_steps[lo:r.nsteps + 1].copy()))
            hits.append((k, irep, int(vit.i_steps[ns]), int(vit.j_steps[ns]), vit.i2, vit.j2, ns, vit.i_steps, vit.j_steps))
            want.append(r)
        tps.append(tp)
        t_lins.append(capi.linear_transitions(ttr, False))
    order = np.random.default_rng(3).permutation(len(hits))
    c = capi.Context()
    sc, re, o_i, o_j, o_s, o_S, o_P = capi.runner_mac_realign(c, qp, q_lin, tps, t_lins, [hits[h] for h in order], loc=local)
    c.close()
    for pos, h in enumerate(order):
        r = want[h]
        assert tuple(sc[pos]) == (r.nsteps, r.i1, r.j1, r.i2, r.j2, r.matched_cols), (h, tuple(sc[pos]))
        assert np.float64(re[pos, 0]).tobytes() == np.float64(r.Pforward).tobytes()
        assert np.float32(re[pos, 1]).tobytes() == np.float32(r.sum_of_probs).tobytes()
        n = r.nsteps
        lo = 0 if n == 0 else 1
        assert np.array_equal(o_i[pos, lo:n + 1], r.i_steps[lo:n + 1]) and np.array_equal(o_j[pos, lo:n + 1], r.j_steps[lo:n + 1])
        assert np.array_equal(o_s[pos, 1:n + 1], r.states[1:n + 1])
        assert o_S[pos, 1:n + 1].tobytes() == r.S[1:n + 1].tobytes() and o_P[pos, 1:n + 1].tobytes() == r.P[1:n + 1].tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("local", [1, 0])
def test_gpu_mac_matches_oracle(oracle, local):
    """All cases as ONE batch (ragged template lengths) per alignment mode, three rounds of alternative alignments."""
    from pyhhv import capi
    cases = [c for c in range(len(CASES)) if CASES[c][0] == 120 or True]
    # one query per launch: group the cases by Lq
    c = capi.Context()
    c.mac_set_lists(1)
    for case in cases:
        qp, qtr, tp, ttr, _, mact = make_pair(case)
        par = make_params(local=local, ss_mode=0)
        vit = oracle.align(par, qp, qtr, tp, ttr, want_path=True)
        q_lin = lin_query(qtr)
        t_lin = lin_template(ttr)
        prev = []
        for rnd in range(3):
            o = oracle_mac_realign(oracle, qp, q_lin, tp, t_lin, vit, local=local, mact=mact, prev=prev)
            ms = c.mac_realign(qp, q_lin, [tp], [t_lin], [o.celloff], local=local, mact=mact)
            h = ms.hits[0]
            assert np.float64(h["Pforward"]).tobytes() == np.float64(o.Pforward).tobytes(), (case, rnd)
            assert ms.posterior(0)[1:, 1:].tobytes() == o.posterior[1:, 1:].tobytes(), (case, rnd)
            got = (h["nsteps"], h["i1"], h["j1"], h["i2"], h["j2"], h["matched_cols"])
            assert got == (o.nsteps, o.i1, o.j1, o.i2, o.j2, o.matched_cols), (case, rnd)
            i_s, j_s, st, S, P = ms.path(0)
            n = o.nsteps
            lo = 0 if n == 0 else 1
            assert np.array_equal(i_s[lo:], o.i_steps[lo:n + 1]) and np.array_equal(j_s[lo:], o.j_steps[lo:n + 1])
            assert np.array_equal(st[1:], o.states[1:n + 1])
            assert S[1:].tobytes() == o.S[1:n + 1].tobytes() and P[1:].tobytes() == o.P[1:n + 1].tobytes()
            assert np.float32(h["sum_of_probs"]).tobytes() == np.float32(o.sum_of_probs).tobytes()
            want = lists_of(o)
            for w in range(3):
                li, lj, lv = ms.list(0, w)
                assert np.array_equal(li, want[w][0]) and np.array_equal(lj, want[w][1]), (case, rnd, w, len(li), len(want[w][0]))
                assert lv.tobytes() == want[w][2].tobytes(), (case, rnd, w)
            assert len(want[0][0]) > 0 and len(want[1][0]) > 0
            ms.free()
            prev.append((o.i_steps[lo:n + 1].copy(), o.j_steps[lo:n + 1].copy()))
    c.close()


@pytest.mark.gpu
def test_gpu_mac_lists_match_reference_in_a_batch(oracle, ref):
    """hhv_mac_set_lists + hhv_mac_list on a ragged batch (all length classes that fit a small test: staged, lean) against the
    lists the reference's writeProfilesToHits attaches to the hit; without hhv_mac_set_lists the forward / backward lists are
    refused and the posterior list still works."""
    from pyhhv import capi
    Lq = 150
    qp, qtr = synth.make_query(777, Lq)
    q_lin = capi.linear_transitions(qtr, True)
    tps, t_lins, cos, want = [], [], [], []
    for k, Lt in enumerate([40, 129, 300, 64, 1100]):
        tp, ttr = synth.make_homolog(40 + k, qp, L=Lt)
        vit = oracle.align(make_params(local=1, ss_mode=0), qp, qtr, tp, ttr, want_path=True)
        r = ref_mac_realign(ref, qp, qtr, tp, ttr, vit, local=1)
        tps.append(tp)
        t_lins.append(r.t_tr_lin)
        cos.append(r.celloff)
        want.append(r)
    c = capi.Context()
    c.mac_set_lists(1)
    ms = c.mac_realign(qp, q_lin, tps, t_lins, cos, local=1)
    for k, r in enumerate(want):
        got = [ms.list(k, w) for w in range(3)]
        same_lists(got, r, Lq)
    ms.free()
    c.mac_set_lists(0)
    ms = c.mac_realign(qp, q_lin, tps, t_lins, cos, local=1)
    with pytest.raises(capi.HhvError):
        ms.list(0, 0)
    got = ms.list(1, 2)
    assert np.array_equal(got[0], want[1].lists[2][0]) and got[2].tobytes() == want[1].lists[2][2].tobytes()
    ms.free()
    c.close()


@pytest.mark.gpu
def test_gpu_device_masks_equal_host_masks(oracle):
    """hhv_mac_realign_hits builds the masks on the device: same bytes as hhv::MacCellOff (pinned to the reference above),
    including excluded cells of earlier alignments and -excl / -template_excl ranges."""
    from pyhhv import capi
    Lq = 130
    qp, qtr = synth.make_query(31, Lq)
    q_lin = capi.linear_transitions(qtr, True)
    tps, t_lins, inputs, want = [], [], [], []
    rng = np.random.default_rng(8)
    for k, Lt in enumerate([50, 127, 128, 260, 5]):
        tp, ttr = synth.make_homolog(300 + k, qp, L=Lt)
        vit = oracle.align(make_params(local=1, ss_mode=0), qp, qtr, tp, ttr, want_path=True)
        ns = vit.nsteps
        nx = int(rng.integers(0, 40))
        xi, xj = rng.integers(1, Lq + 1, nx).astype(np.int32), rng.integers(1, Lt + 1, nx).astype(np.int32)
        hit = (int(vit.i_steps[ns]), int(vit.j_steps[ns]), vit.i2, vit.j2, ns, vit.i_steps, vit.j_steps)
        want.append(capi.mac_celloff(Lq, Lt, hit, [(xi, xj)] if nx else [], exclstr="10-20,100-400", template_exclstr="3-4"))
        inputs.append(hit + (xi, xj))
        tps.append(tp)
        t_lins.append(capi.linear_transitions(ttr, False))
    c = capi.Context()
    ms = c.mac_realign_hits(qp, q_lin, tps, t_lins, inputs, qranges=[10, 20, 100, 400], tranges=[3, 4])
    for k in range(len(tps)):
        assert np.array_equal(ms.celloff(k), want[k]), k
    ms.free()
    c.close()


@pytest.mark.gpu
def test_gpu_mac_from_resident_template_set(oracle):
    """hhv_mac_realign_tset reads the template profiles from the record stream the Viterbi stage searched: same results as
    staging the profiles from the host, for hits that reference the resident set out of order and twice."""
    from pyhhv import capi
    Lq = 120
    qp, qtr = synth.make_query(555, Lq)
    q_lin = capi.linear_transitions(qtr, True)
    tps, ttrs = zip(*[synth.make_homolog(700 + k, qp, L=L) for k, L in enumerate([60, 200, 33, 128, 90])])
    c = capi.Context(local=1, ss_mode=0)
    c.set_query(qp, qtr)
    ts = c.upload(list(tps), list(ttrs))
    c.align(ts, backtrace=True)
    vh = c.hits(ts)
    tof = [3, 0, 4, 0, 1]
    inputs, t_lins = [], []
    for t in tof:
        ns, i_s, j_s, st, S = c.hit_path(ts, t)
        inputs.append((int(vh["i1"][t]), int(vh["j1"][t]), int(vh["i2"][t]), int(vh["j2"][t]), ns, i_s, j_s,
                       np.zeros(0, np.int32), np.zeros(0, np.int32)))
        t_lins.append(capi.linear_transitions(ttrs[t], False))
    a = c.mac_realign_tset(qp, q_lin, ts, tof, t_lins, inputs)
    b = c.mac_realign_hits(qp, q_lin, [tps[t] for t in tof], t_lins, inputs)
    assert a.hits.tobytes() == b.hits.tobytes()
    for k in range(len(tof)):
        assert a.posterior(k).tobytes() == b.posterior(k).tobytes()
        for x, y in zip(a.path(k), b.path(k)):
            assert np.array_equal(x, y)
    a.free()
    b.free()
    ts.free()
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lengths", [[900, 70, 1500], [2100, 64, 810, 2047, 2046, 3000, 300, 500, 790, 200]])
def test_gpu_mac_length_classes(oracle, lengths):
    """The hits of a call are launched by length class: template and row state in LDS (up to ~800 columns), row state in LDS
    and the template read from global memory (up to 2046), row state in global memory too (any length).  Results must not
    depend on the class, and one long template must not change what the others get."""
    from pyhhv import capi
    Lq = 90 if max(lengths) < 2000 else 48
    qp, qtr = synth.make_query(88, Lq)
    q_lin = lin_query(qtr)
    par = make_params(local=1, ss_mode=0)
    tps, tls, masks, want = [], [], [], []
    for k, Lt in enumerate(lengths):
        tp, ttr = synth.make_homolog(800 + k, qp, L=Lt)
        t_lin = lin_template(ttr)
        vit = oracle.align(par, qp, qtr, tp, ttr, want_path=True)
        o = oracle_mac_realign(oracle, qp, q_lin, tp, t_lin, vit, local=1)
        tps.append(tp)
        tls.append(t_lin)
        masks.append(o.celloff)
        want.append(o)
    c = capi.Context()
    ms = c.mac_realign(qp, q_lin, tps, tls, masks, local=1)
    for k, o in enumerate(want):
        h = ms.hits[k]
        assert np.float64(h["Pforward"]).tobytes() == np.float64(o.Pforward).tobytes(), k
        assert ms.posterior(k)[1:, 1:].tobytes() == o.posterior[1:, 1:].tobytes(), k
        assert (h["nsteps"], h["i1"], h["j1"], h["i2"], h["j2"]) == (o.nsteps, o.i1, o.j1, o.i2, o.j2), k
    ms.free()
    c.close()


@pytest.mark.gpu
def test_gpu_mac_more_hits_than_resident_workgroups(oracle):
    """The dataflow classes take the longest hits of a batch up to what the GPU holds at once (hhv_api_mac.cpp /
    mac_dataflow_budget), the shortest ones beyond it go to the single-wave kernels of the class without LDS: 1 400 hits of
    20-40 columns together with a few long ones that stay in their dataflow classes; results must not depend on the class taken."""
    from pyhhv import capi
    Lq = 36
    qp, qtr = synth.make_query(91, Lq)
    q_lin = lin_query(qtr)
    par = make_params(local=1, ss_mode=0)
    base = []
    for k, Lt in enumerate([20, 24, 31, 33, 40, 28, 300, 200]):
        tp, ttr = synth.make_homolog(1700 + k, qp, L=Lt)
        t_lin = lin_template(ttr)
        vit = oracle.align(par, qp, qtr, tp, ttr, want_path=True)
        base.append((tp, t_lin, oracle_mac_realign(oracle, qp, q_lin, tp, t_lin, vit, local=1)))
    idx = [k % 6 for k in range(1400)] + [6, 7, 6]
    c = capi.Context()
    ms = c.mac_realign(qp, q_lin, [base[i][0] for i in idx], [base[i][1] for i in idx], [base[i][2].celloff for i in idx], local=1)
    for e, i in enumerate(idx):
        o, h = base[i][2], ms.hits[e]
        assert np.float64(h["Pforward"]).tobytes() == np.float64(o.Pforward).tobytes(), e
        assert (h["nsteps"], h["i1"], h["j1"], h["i2"], h["j2"]) == (o.nsteps, o.i1, o.j1, o.i2, o.j2), e
    for e in (0, 5, 700, 1399, 1400, 1402):
        assert ms.posterior(e)[1:, 1:].tobytes() == base[idx[e]][2].posterior[1:, 1:].tobytes(), e
    ms.free()
    # one level further: a lean class (row state in LDS: two workgroups per CU at 1 000 columns) with more hits than that gives
    # them to the class without LDS
    long_ = []
    for k, Lt in enumerate([1000, 990]):
        tp, ttr = synth.make_homolog(1800 + k, qp, L=Lt)
        t_lin = lin_template(ttr)
        vit = oracle.align(par, qp, qtr, tp, ttr, want_path=True)
        long_.append((tp, t_lin, oracle_mac_realign(oracle, qp, q_lin, tp, t_lin, vit, local=1)))
    idx = [k % 2 for k in range(620)]
    ms = c.mac_realign(qp, q_lin, [long_[i][0] for i in idx], [long_[i][1] for i in idx], [long_[i][2].celloff for i in idx], local=1)
    for e, i in enumerate(idx):
        o, h = long_[i][2], ms.hits[e]
        assert np.float64(h["Pforward"]).tobytes() == np.float64(o.Pforward).tobytes(), e
        assert (h["nsteps"], h["i1"], h["j1"], h["i2"], h["j2"]) == (o.nsteps, o.i1, o.j1, o.i2, o.j2), e
    for e in (0, 1, 619):
        assert ms.posterior(e)[1:, 1:].tobytes() == long_[idx[e]][2].posterior[1:, 1:].tobytes(), e
    ms.free()
    c.close()


@pytest.mark.gpu
def test_gpu_mac_batch_of_ragged_hits(oracle):
    """Many hits of one query in one launch (ragged Lt, some without mask) equal the same hits done one by one."""
    from pyhhv import capi
    Lq = 150
    qp, qtr = synth.make_query(77, Lq)
    q_lin = lin_query(qtr)
    par = make_params(local=1, ss_mode=0)
    tps, tls, masks, want = [], [], [], []
    for k, Lt in enumerate([40, 64, 65, 128, 129, 200, 333, 1, 2, 3]):
        tp, ttr = synth.make_homolog(500 + k, qp, L=Lt) if k % 3 else synth.make_template(500 + k, Lt)
        t_lin = lin_template(ttr)
        vit = oracle.align(par, qp, qtr, tp, ttr, want_path=True)
        o = oracle_mac_realign(oracle, qp, q_lin, tp, t_lin, vit, local=1)
        if k % 4 == 3:   # no mask at all
            o = oracle_mac_realign_nomask(oracle, qp, q_lin, tp, t_lin)
            masks.append(None)
        else:
            masks.append(o.celloff)
        tps.append(tp)
        tls.append(t_lin)
        want.append(o)
    c = capi.Context()
    ms = c.mac_realign(qp, q_lin, tps, tls, masks, local=1)
    for k, o in enumerate(want):
        h = ms.hits[k]
        assert np.float64(h["Pforward"]).tobytes() == np.float64(o.Pforward).tobytes(), k
        assert ms.posterior(k)[1:, 1:].tobytes() == o.posterior[1:, 1:].tobytes(), k
        assert (h["nsteps"], h["i1"], h["j1"], h["i2"], h["j2"]) == (o.nsteps, o.i1, o.j1, o.i2, o.j2), k
        i_s, j_s, st, S, P = ms.path(k)
        assert P[1:].tobytes() == o.P[1:o.nsteps + 1].tobytes() and np.array_equal(st[1:], o.states[1:o.nsteps + 1])
    ms.free()
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("local", [1, 0])
def test_gpu_mac_ring_and_wide_rows(oracle, local):
    """Templates beyond the plain LDS layout (> ~1450 columns) run the dataflow kernels on a RING of 22 strips when every row's
    visited span fits it, the single-wave kernels otherwise - decided on the device from the masks (hhv_mac.hip MAC_RING_STRIPS,
    rng[0].x).  One batch with both kinds: band masks around a Viterbi path (ring), an all-on mask and a mask with a wide block
    left of the alignment (rows of 27 and more strips: single-wave), a band that wraps the ring several times (3 000 columns),
    next to short templates of the plain classes; local and global.  Everything bit for bit against the oracle."""
    from pyhhv import capi
    Lq = 70
    qp, qtr = synth.make_query(188, Lq)
    q_lin = lin_query(qtr)
    par = make_params(local=local, ss_mode=0)
    lengths = [1700, 1700, 2500, 3000, 1700, 300, 1460, 1456]
    tps, tls, masks, want = [], [], [], []
    for k, Lt in enumerate(lengths):
        tp, ttr = synth.make_homolog(900 + k, qp, L=Lt)
        t_lin = lin_template(ttr)
        if k == 0:      # every cell on: rows 27 strips wide
            o = oracle_mac_realign_nomask(oracle, qp, q_lin, tp, t_lin, local=local)
        else:
            vit = oracle.align(par, qp, qtr, tp, ttr, want_path=True)
            o = oracle_mac_realign(oracle, qp, q_lin, tp, t_lin, vit, local=local)
        tps.append(tp)
        tls.append(t_lin)
        masks.append(o.celloff)
        want.append(o)
    # hit 4: the band of hit 1's kind plus a block of active cells far from it in a few rows (a wide row in the middle of narrow ones)
    m = masks[4].copy()
    m[20:24, 1:1650] = 0
    masks[4] = m

    class V:
        pass
    c = capi.Context()
    ms = c.mac_realign(qp, q_lin, tps, tls, masks, local=local)
    for k in range(len(lengths)):
        if k == 4:
            continue
        o, h = want[k], ms.hits[k]
        assert np.float64(h["Pforward"]).tobytes() == np.float64(o.Pforward).tobytes(), k
        assert ms.posterior(k)[1:, 1:].tobytes() == o.posterior[1:, 1:].tobytes(), k
        assert (h["nsteps"], h["i1"], h["j1"], h["i2"], h["j2"]) == (o.nsteps, o.i1, o.j1, o.i2, o.j2), k
    # hit 4 against the same hit run alone with the ring switched off by its width: determinism across the two kinds is what the
    # others check against the oracle; here the widened mask has no oracle result, so compare with a second context's run
    ms2 = c.mac_realign(qp, q_lin, [tps[4]], [tls[4]], [masks[4]], local=local)
    assert ms.posterior(4).tobytes() == ms2.posterior(0).tobytes()
    assert np.float64(ms.hits[4]["Pforward"]).tobytes() == np.float64(ms2.hits[0]["Pforward"]).tobytes()
    ms2.free()
    ms.free()
    c.close()
