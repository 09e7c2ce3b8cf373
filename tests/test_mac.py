"""MAC realignment: PosteriorDecoder::realign = forward / backward / MAC DP / MAC backtrace
(/root/reference src/hhposteriordecoder.cpp:86-119 and the four algorithm files).
CPU: the oracle restatement against the reference's own member functions (oracle/ref_mac_harness.cpp), bit for bit.
GPU: the HIP kernels against the oracle."""
import numpy as np
import pytest

from pyhhv import synth
from pyoracle import make_params, oracle_mac_realign, ref_mac_realign

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


def same(a, b):
    assert a.nsteps == b.nsteps and (a.i1, a.j1, a.i2, a.j2, a.matched_cols) == (b.i1, b.j1, b.i2, b.j2, b.matched_cols)
    n = a.nsteps
    lo = 0 if n == 0 else 1   # entry 0 is only written when the backtrace does not start in a match state
    assert np.array_equal(a.i_steps[lo:n + 1], b.i_steps[lo:n + 1]) and np.array_equal(a.j_steps[lo:n + 1], b.j_steps[lo:n + 1])
    assert np.array_equal(a.states[1:n + 1], b.states[1:n + 1])
    assert a.S[1:n + 1].tobytes() == b.S[1:n + 1].tobytes() and a.P[1:n + 1].tobytes() == b.P[1:n + 1].tobytes()
    assert np.float32(a.sum_of_probs).tobytes() == np.float32(b.sum_of_probs).tobytes()


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
        prev.append((r.i_steps[(0 if r.nsteps == 0 else 1):r.nsteps + 1].copy(), r.j_steps[(0 if r.nsteps == 0 else 1):r.nsteps + 1].copy()))
    assert r.nsteps >= 0
