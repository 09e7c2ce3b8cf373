"""Pin the CPU restatement (oracle/hhv_oracle.c) to the REFERENCE ITSELF (oracle/_ref/libhhref.so =
the reference's own translation units compiled by oracle/Makefile).  Everything is bit-exact."""
import numpy as np
import pytest

from common import workload
from pyoracle import make_params


def bits(x):
    return np.float32(x).tobytes()


def test_log2f4_bitexact(oracle, ref):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.random(4000).astype(np.float32) * 4,
                         np.array([0, 1, 2, 0.5, 1e-30, 3e38, 1e-45], dtype=np.float32),
                         (2.0 ** rng.uniform(-30, 30, 4000)).astype(np.float32)])
    for x in xs:
        assert bits(oracle.log2f4(float(x))) == bits(ref.log2f4(float(x))), x
    assert oracle.log2f4(0.0) == -127.0


def test_fast_log2_table(oracle, ref):
    for b in range(1024):
        for c in (0, 1, 4097, 8191):
            x = float(np.uint32(0x3F800000 | (b << 13) | c).view(np.float32))
            assert bits(oracle.fast_log2(x)) == bits(ref.fast_log2(x)), (b, c)
    for x in (0.0, -1.0, 1e-30, 3.7, 1e20):
        assert bits(oracle.fast_log2(x)) == bits(ref.fast_log2(x))


def test_dot_products(oracle, ref):
    rng = np.random.default_rng(2)
    for _ in range(1500):
        q = rng.random(20).astype(np.float32)
        t = (rng.random(20) ** 3 * 5).astype(np.float32)
        assert bits(oracle.dot20_vec(q, t)) == bits(ref.dot20_vec(q, t))
        assert bits(oracle.dot20_scalar(q, t)) == bits(ref.dot20_scalar(q, t))


@pytest.mark.parametrize("case", range(12))
def test_align_backtrace_score_bitexact(oracle, ref, case):
    """Mixed-length SIMD batches (incl. the global-mode batch-composition quirk):
    ViterbiResult, every backtrace byte, the path, per-column S and the Hit score."""
    rng = np.random.default_rng(100 + case)
    Lq = int(rng.integers(20, 140))
    par = make_params(local=case % 2, egq=0.0 if case % 4 < 2 else 0.3, egt=0.0 if case % 4 < 2 else 0.1)
    n = int(rng.integers(1, ref.V + 1))
    qf, qtr, tps, ttrs = workload(case, Lq, n, 10, 160)
    outs = ref.align_batch(par, qf, qtr, tps, ttrs, want_path=True)
    Ltb = max(p.shape[0] - 1 for p in tps)
    for e in range(n):
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], Lbatch=Ltb, want_path=True)
        b = outs[e]
        assert bits(a.score) == bits(b.score) and (a.i2, a.j2) == (b.i2, b.j2)
        assert np.array_equal(a.bt[1:, 1:], b.bt[1:, 1:])
        ns = a.nsteps
        assert (ns, a.matched_cols) == (b.nsteps, b.matched_cols)
        assert np.array_equal(a.i_steps[1:ns + 1], b.i_steps[1:ns + 1])
        assert np.array_equal(a.j_steps[1:ns + 1], b.j_steps[1:ns + 1])
        assert np.array_equal(a.states[1:ns + 1], b.states[1:ns + 1])
        assert np.array_equal(a.S[1:ns + 1].view(np.uint32), b.S[1:ns + 1].view(np.uint32))
        assert bits(a.hit_score) == bits(b.hit_score)


def test_single_length_batch_equals_native(oracle, ref):
    """MapOneHMM batches (the definition of the oracle for mixed-length configs, SURVEY.md 8d cfg 5)."""
    par = make_params(local=0)
    qf, qtr, tps, ttrs = workload(77, 90, 6, 30, 120)
    for e in range(6):
        b = ref.align_batch(par, qf, qtr, [tps[e]], [ttrs[e]], replicate=True, want_path=True)[0]
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_path=True)
        assert bits(a.score) == bits(b.score) and (a.i2, a.j2) == (b.i2, b.j2)
        assert np.array_equal(a.bt[1:, 1:], b.bt[1:, 1:])
        assert bits(a.hit_score) == bits(b.hit_score)


def test_exclude_alignment_and_celloff(oracle, ref):
    """Alt-alignment round 2: ExcludeAlignment mask + AlignWithCellOff, both modes."""
    for local in (0, 1):
        par = make_params(local=local)
        qf, qtr, tps, ttrs = workload(31 + local, 110, 4, 60, 140, homolog_every=1)
        first = ref.align_batch(par, qf, qtr, tps, ttrs, want_path=True)
        masks = []
        for e in range(4):
            Lt = tps[e].shape[0] - 1
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], Lbatch=max(p.shape[0] - 1 for p in tps), want_path=True)
            m_ref = ref.exclude_alignment(110, Lt, first[e].i_steps, first[e].j_steps, first[e].nsteps)
            m_or = oracle.exclude_alignment(110, Lt, a.i_steps, a.j_steps, a.nsteps)
            assert np.array_equal(m_ref, m_or)
            assert m_ref.sum() > 0
            masks.append(m_ref)
        second = ref.align_batch(par, qf, qtr, tps, ttrs, celloffs=masks, want_path=True)
        Ltb = max(p.shape[0] - 1 for p in tps)
        changed = 0
        for e in range(4):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], Lbatch=Ltb, celloff=masks[e], want_path=True)
            b = second[e]
            assert bits(a.score) == bits(b.score) and (a.i2, a.j2) == (b.i2, b.j2)
            assert np.array_equal(a.bt[1:, 1:], b.bt[1:, 1:])
            assert bits(a.hit_score) == bits(b.hit_score)
            changed += int(bits(b.score) != bits(first[e].score))
        assert changed > 0


def test_secondary_structure_variant(oracle, ref):
    """...AndSS kernels for the three ss_hmm_modes."""
    from pyoracle import SSInfo
    rng = np.random.default_rng(5)
    S73 = rng.normal(0, 1, (8, 4, 11)).astype(np.float32)
    S33 = rng.normal(0, 1, (4, 11, 4, 11)).astype(np.float32)
    S37 = rng.normal(0, 1, (4, 11, 8)).astype(np.float32)
    par = make_params(local=1, ss_mode=2)
    qf, qtr, tps, ttrs = workload(9, 80, 3, 40, 100, homolog_every=1)
    Lq = 80
    for mode in (1, 2, 4):
        ss = SSInfo(mode, rng.integers(0, 4, Lq + 1), rng.integers(0, 11, Lq + 1), rng.integers(0, 8, Lq + 1), S73,
                    S33, S37)
        t_sss = [(rng.integers(0, 4, p.shape[0]), rng.integers(0, 11, p.shape[0]), rng.integers(0, 8, p.shape[0]))
                 for p in tps]
        outs = ref.align_batch(par, qf, qtr, tps, ttrs, ss=ss, t_sss=t_sss, want_path=True)
        Ltb = max(p.shape[0] - 1 for p in tps)
        for e in range(3):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], Lbatch=Ltb, ss=ss, t_ss=t_sss[e], want_path=True)
            b = outs[e]
            assert bits(a.score) == bits(b.score) and (a.i2, a.j2) == (b.i2, b.j2), mode
            assert np.array_equal(a.bt[1:, 1:], b.bt[1:, 1:])
            assert bits(a.hit_score) == bits(b.hit_score) and bits(a.score_ss) == bits(b.score_ss)


def test_bench_hits_oracle_equals_reference(oracle, ref):
    """The batch loops with backtrace used by the configs[2] / configs[4] parity tests and by bench.py: restatement ==
    reference for every output incl. the path checksums, batched (equal lengths) and replicated (ragged lengths)."""
    from pyhhv import synth
    from pyoracle import path_hashes
    qf, qtr = synth.make_query(5, 90)
    for local, ragged in ((0, False), (1, True), (0, True)):
        par = make_params(local=local)
        tps, ttrs = [], []
        for k in range(37):
            L = 70 + (k * 7) % 50 if ragged else 80
            p, tr = synth.make_homolog(900 + k, qf, L=L) if k % 2 else synth.make_template(900 + k, L)
            tps.append(p)
            ttrs.append(tr)
        a = ref.bench_hits(par, qf, qtr, tps, ttrs, threads=3, replicate=ragged)
        b = oracle.bench_hits(par, qf, qtr, tps, ttrs, threads=3)
        for k in a:
            if k != "sec":
                assert np.array_equal(a[k], b[k]), (local, ragged, k)
        # the numpy restatement of the checksums (what the GPU tests apply to the engine's path pool)
        o = oracle.align(par, qf, qtr, tps[3], ttrs[3], want_path=True)
        ph, sh = path_hashes([0], o.i_steps, o.j_steps, o.states, o.S, [o.nsteps])
        assert ph[0] == a["path_hash"][3] and sh[0] == a["s_hash"][3]
