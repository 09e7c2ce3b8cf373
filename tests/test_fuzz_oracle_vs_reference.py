"""CPU fuzz: the oracle restatements against the reference's own code (oracle/_ref) on adversarial inputs the golden
cases lack - exact ties (scores on a coarse grid), -100000 transitions in the middle of a profile, lengths 1-3, random
cell-off masks, odd pseudocount / gap parameters.  The GPU soak (tools/soak.py) compares the kernels with the oracle on
the same families, so together they tie the kernels to the reference."""
import numpy as np
import pytest

import pyoracle as po
from pyhhv import synth


def quant(a):
    b = (np.round(a / 0.5) * 0.5).astype(np.float32)
    b[b < -1000] = -100000.0
    return b


@pytest.mark.parametrize("seed", range(4))
def test_viterbi_ties_masks_tiny_lengths(oracle, ref, seed):
    rng = np.random.default_rng(9000 + seed)
    for _ in range(40):
        Lq = int(rng.choice([1, 2, 5, 40, 65, 130]))
        local = int(rng.integers(0, 2))
        par = po.make_params(local=local, egq=float(rng.choice([0.0, 0.2])), egt=float(rng.choice([0.0, 0.1])),
                             shift=float(rng.choice([-0.03, 0.0, 0.25])), ss_mode=0)
        qp, qtr = synth.make_query(int(rng.integers(1 << 30)), Lq)
        if rng.random() < 0.5:
            qtr = quant(qtr)
        Lt = int(rng.choice([1, 2, 3, 31, 64, 130]))
        tp, ttr = (synth.make_homolog(int(rng.integers(1 << 30)), qp, L=Lt) if rng.random() < 0.5 and Lq > 4 else
                   synth.make_template(int(rng.integers(1 << 30)), Lt))
        if rng.random() < 0.5:
            ttr = quant(ttr)
        if rng.random() < 0.2:
            ttr[rng.integers(0, Lt + 1), rng.integers(0, 7)] = -100000.0
        m = (rng.random((Lq + 1, Lt + 1)) < rng.choice([0.0, 0.05, 0.5])).astype(np.uint8) if rng.random() < 0.5 else None
        a = oracle.align(par, qp, qtr, tp, ttr, celloff=m, want_path=True)
        r = ref.align_batch(par, qp, qtr, [tp], [ttr], celloffs=[m] if m is not None else None, want_path=True)[0]
        assert (a.i2, a.j2, a.nsteps) == (r.i2, r.j2, r.nsteps)
        assert np.float32(a.score).tobytes() == np.float32(r.score).tobytes()
        assert np.array_equal(a.bt[1:, 1:] & 0x7F, r.bt[1:, 1:Lt + 1] & 0x7F)
        assert np.float32(a.hit_score).tobytes() == np.float32(r.hit_score).tobytes()


@pytest.mark.parametrize("seed", range(3))
def test_prepare_odd_parameters(oracle, ref, seed):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gonnet_pb_R.npz"))
    pb, R = z["pb"], z["R"]
    rng = np.random.default_rng(9100 + seed)
    for _ in range(40):
        L = int(rng.choice([1, 2, 63, 64, 65, 255, 300]))
        f, tr, neff, nh = synth.make_raw_hmm(int(rng.integers(1 << 30)), L)
        if rng.random() < 0.3:
            j = int(rng.integers(1, L + 1))
            f[j] = 2.0 ** -99.999
            f[j, rng.integers(0, 20)] = 1.0
            tr[j - 1] = [0.0, -99.999, -99.999, 0.0, -99.999, 0.0, -99.999]
        if rng.random() < 0.2:
            neff[:, 0] = 1.0
        pc = np.array([int(rng.integers(0, 3)), float(rng.choice([1.0, 0.4, 0.0, 1.7])), float(rng.choice([1.5, 0.5, 4.0])), 1.0],
                      np.float32)
        gap = np.array([rng.choice([0.15, 1.0]), rng.choice([1.0, 0.3, 2.0]), 0.6, rng.choice([0.6, 1.0]), 0.6, 0.6,
                        rng.choice([1.0, 0.0, 2.5])], np.float32)
        cs = int(rng.integers(0, 4))
        q_pav = rng.dirichlet(np.ones(20) * 5).astype(np.float32)
        for role in (0, 1):
            a = po.ref_prepare(ref, role, f, tr, neff, nh, q_pav=q_pav, gap=gap, pc=pc, columnscore=cs, pb=pb)
            b = po.oracle_prepare(oracle, role, f, tr, neff, nh, pb, R, q_pav=q_pav, gap=gap, pc=pc, columnscore=cs)
            for x, y in zip(a, b):
                assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (L, pc, gap, cs, role)
