"""HHblits prefilter kernels: Prefilter::ungapped_sse_score and Prefilter::swStripedByte
(/root/reference src/hhprefilter.cpp:214-278, 70-212).
CPU: the oracle restatements are pinned to the reference's own functions (oracle/ref_prefilter_harness.cpp).
GPU: the HIP kernels (hhv_prefilter_scores) against the oracle."""
import ctypes as C

import numpy as np
import pytest

OFFSET, GAP_INIT, GAP_EXT = 50, 24, 4   # prefilter_score_offset, gap_open + gap_extend, gap_extend (src/hhdecl.cpp:120-123)


def u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def make_case(seed, n=16, strong=True):
    """Random 220-state byte profile + db sequences, every second one a noisy, gapped copy of the consensus."""
    rng = np.random.default_rng(seed)
    Lq = int(rng.integers(5, 330))
    prof = np.clip(rng.normal(25 if strong else 38, 12, (220, Lq)), 0, 255).astype(np.uint8)
    cons = rng.integers(0, 220, Lq)
    if strong:
        prof[cons, np.arange(Lq)] = rng.integers(60, 80, Lq)
    lens = rng.integers(1, 450, n)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    seqs = rng.integers(0, 220, offs[-1]).astype(np.uint8)
    for k in range(0, n, 2):
        s, L = offs[k], lens[k]
        qpos, t = int(rng.integers(0, max(1, Lq // 3))), 0
        while t < L and qpos < Lq:
            r = rng.random()
            if r < 0.05:
                qpos += int(rng.integers(1, 4))
                continue
            if r < 0.10:
                t += 1
                continue
            if rng.random() < 0.9:
                seqs[s + t] = cons[qpos]
            t += 1
            qpos += 1
    return prof, Lq, seqs, offs


def oracle_scores(orc, prof, Lq, seqs, offs):
    n = len(offs) - 1
    ung = np.zeros(n, dtype=np.int32)
    gap = np.zeros(n, dtype=np.int32)
    for k in range(n):
        s = np.ascontiguousarray(seqs[offs[k]:offs[k + 1]])
        ung[k] = orc.lib.hho_ungapped_score(u8(prof), Lq, u8(s), len(s), OFFSET)
        gap[k] = orc.lib.hho_sw_score(u8(prof), Lq, u8(s), len(s), GAP_INIT, GAP_EXT, OFFSET, 32)
    return ung, gap


@pytest.mark.parametrize("seed", range(10))
def test_oracle_prefilter_kernels_match_reference(oracle, ref, seed):
    prof, Lq, seqs, offs = make_case(seed, strong=seed % 3 != 0)
    n = len(offs) - 1
    ung = np.zeros(n, dtype=np.int32)
    gap = np.zeros(n, dtype=np.int32)
    assert ref.lib.ref_prefilter_vecbytes() == 32
    ref.lib.ref_prefilter_scores(u8(prof), Lq, u8(seqs), offs.ctypes.data_as(C.POINTER(C.c_long)), n, OFFSET, GAP_INIT,
                                 GAP_EXT, ung.ctypes.data_as(C.POINTER(C.c_int)), gap.ctypes.data_as(C.POINTER(C.c_int)))
    o_ung, o_gap = oracle_scores(oracle, prof, Lq, seqs, offs)
    assert np.array_equal(ung, o_ung) and np.array_equal(gap, o_gap)
    assert gap.max() >= ung.max() >= 0 and gap.max() <= 255 - OFFSET


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_gpu_prefilter_scores_match_oracle(oracle, seed):
    from pyhhv import capi
    prof, Lq, seqs, offs = make_case(100 + seed, n=200, strong=seed % 3 != 0)
    o_ung, o_gap = oracle_scores(oracle, prof, Lq, seqs, offs)
    c = capi.Context()
    db = c.prefilter_upload_db(seqs, offs)
    ung = c.prefilter_scores(db, prof, OFFSET, gapped=False)
    assert np.array_equal(ung, o_ung)
    subset = np.arange(0, len(offs) - 1, 3, dtype=np.int32)
    gap = c.prefilter_scores(db, prof, OFFSET, gapped=True, gap_init=GAP_INIT, gap_extend=GAP_EXT, subset=subset)
    assert np.array_equal(gap, o_gap[subset])
    c.prefilter_free_db(db)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("Lq", [513, 640, 700, 1100, 2049, 6816, 6817, 9001])
def test_gpu_prefilter_long_queries(oracle, Lq, monkeypatch):
    """Queries beyond one slab: the gapless kernel runs in slabs of <= 512 rows with carried diagonals, Smith-Waterman
    falls back to the generic kernel (striped profile in LDS up to Lq = 640, read through L2 beyond; its H/E columns in LDS
    up to Lq = 6816, in global memory beyond - no query length is refused)."""
    from pyhhv import capi
    if Lq > 6000:
        monkeypatch.setenv("HHV_PREFILTER_GENERIC", "1")    # the gapless scores through the generic kernel as well
    rng = np.random.default_rng(Lq)
    prof = np.clip(rng.normal(30, 12, (220, Lq)), 0, 255).astype(np.uint8)
    cons = rng.integers(0, 220, Lq)
    prof[cons, np.arange(Lq)] = rng.integers(56, 64, Lq)
    n = 60
    lens = rng.integers(1, 900, n)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    seqs = rng.integers(0, 220, offs[-1]).astype(np.uint8)
    for k in range(0, n, 2):     # every second sequence follows a stretch of the consensus that crosses slab borders
        q0 = int(rng.integers(max(0, 512 - 200), min(Lq - 1, 512 + 50))) if Lq > 600 else int(rng.integers(0, Lq // 2))
        L = min(int(lens[k]), Lq - q0)
        keep = rng.random(L) < 0.85
        seqs[offs[k]:offs[k] + L][keep] = cons[q0:q0 + L][keep]
    o_ung, o_gap = oracle_scores(oracle, prof, Lq, seqs, offs)
    c = capi.Context()
    db = c.prefilter_upload_db(seqs, offs)
    assert np.array_equal(c.prefilter_scores(db, prof, OFFSET, gapped=False), o_ung)
    sub = np.arange(1, n, 2, dtype=np.int32)[::-1].copy()
    assert np.array_equal(c.prefilter_scores(db, prof, OFFSET, gapped=False, subset=sub), o_ung[sub])
    assert np.array_equal(c.prefilter_scores(db, prof, OFFSET, gapped=True, gap_init=GAP_INIT, gap_extend=GAP_EXT), o_gap)
    assert o_ung.max() > 100
    c.prefilter_free_db(db)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_gpu_sw_on_homologs_for_any_gap_parameters(oracle, seed):
    """The sequences Smith-Waterman really sees - survivors of the gapless stage: long stretches that follow the profile's best
    states, scores up to the cap - under many gap parameters and query lengths (1 .. 16 cells per stripe element).  This is where
    the lazy-F correction works hardest (40-60 rows behind a high-scoring cell); the kernel computes it as a prefix scan when
    gap_init >= gap_extend and with the reference's loop otherwise (hhv_pf_sw_kernel<W, SCAN>): both against the oracle, which is
    pinned to Prefilter::swStripedByte for any gap parameters (test_oracle_sw_matches_reference_for_any_gap_parameters)."""
    from pyhhv import capi
    rng = np.random.default_rng(4200 + seed)
    # (up to 640 columns: hhv_pf_sw_kernel, H and E in registers; beyond: the generic kernel with its columns in LDS - both have the scan)
    Lq = (7, 33, 64, 100, 160, 257, 300, 320, 448, 512, 700, 1100)[seed]
    off = int(rng.choice([0, 20, 50]))
    prof = np.clip(rng.normal(off - 6, 10, (220, Lq)), 0, 255).astype(np.uint8)
    best = rng.integers(0, 219, Lq)
    prof[best, np.arange(Lq)] = np.clip(off + rng.integers(3, 30, Lq), 0, 255)
    n = 240
    lens = rng.integers(1, 700, n)
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    seqs = rng.integers(0, 219, offs[-1]).astype(np.uint8)
    for k in range(n):
        if k % 4 == 3:
            continue                      # (a quarter stays random)
        L, q, t = int(lens[k]), int(rng.integers(0, max(1, Lq // 2))), 0
        frac = (0.3, 0.7, 0.95)[k % 3]
        while t < L:
            r = rng.random()
            if r < 0.03:
                q += int(rng.integers(1, 12))     # a deletion: the vertical gaps the lazy-F correction is about
            elif r < 0.06:
                t += int(rng.integers(1, 6))
                continue
            if q >= Lq:
                q = int(rng.integers(0, max(1, Lq // 2)))   # a second copy (repeat)
            if rng.random() < frac:
                seqs[offs[k] + t] = best[q]
            t += 1
            q += 1
    c = capi.Context()
    db = c.prefilter_upload_db(seqs, offs)
    seen_cap = False
    for go, ge in ((24, 4), (20, 4), (4, 4), (9, 9), (60, 1), (5, 0), (0, 0), (3, 4), (0, 9), (30, 30)):
        want = np.array([oracle.lib.hho_sw_score(u8(prof), Lq, u8(np.ascontiguousarray(seqs[offs[k]:offs[k + 1]])), int(lens[k]), go, ge,
                                                 off, 32) for k in range(n)], dtype=np.int32)
        got = c.prefilter_scores(db, prof, off, gapped=True, gap_init=go, gap_extend=ge)
        assert np.array_equal(got, want), (Lq, off, go, ge, int((got != want).sum()), np.flatnonzero(got != want)[:5])
        seen_cap = seen_cap or int(want.max()) == 255 - off
    assert seen_cap or Lq < 31
    c.prefilter_free_db(db)
    c.close()


@pytest.mark.parametrize("seed", range(6))
def test_oracle_sw_matches_reference_for_any_gap_parameters(oracle, ref, seed):
    """gap_init < gap_extend never happens in hhblits (gap_init = open + extend), but the striped algorithm is defined
    for it: its lazy-F exit test then prunes the F chain, and the restatement must follow it literally."""
    rng = np.random.default_rng(900 + seed)
    for _ in range(40):
        Lq = int(rng.choice([1, 31, 33, 64, 65, 100, 257]))
        off = int(rng.choice([0, 20, 50]))
        prof = np.clip(rng.normal(off - 4, 14, (220, Lq)), 0, 255).astype(np.uint8)
        L = int(rng.integers(1, 40))
        seq = rng.integers(0, 220, L).astype(np.uint8)
        go, ge = int(rng.choice([0, 3, 5, 24, 60])), int(rng.choice([0, 1, 4, 9]))
        offs = np.array([0, L], np.int64)
        gap = np.zeros(1, np.int32)
        ung = np.zeros(1, np.int32)
        ref.lib.ref_prefilter_scores(u8(prof), Lq, u8(seq), offs.ctypes.data_as(C.POINTER(C.c_long)), 1, off, go, ge,
                                     ung.ctypes.data_as(C.POINTER(C.c_int)), gap.ctypes.data_as(C.POINTER(C.c_int)))
        assert oracle.lib.hho_sw_score(u8(prof), Lq, u8(seq), L, go, ge, off, 32) == gap[0], (Lq, off, L, go, ge)
        assert oracle.lib.hho_ungapped_score(u8(prof), Lq, u8(seq), L, off) == ung[0]


# ---- host side: flog2 / fpow2, context library, query profile, the two selection steps of prefilter_db ------------
import os

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_CS219 = "/root/reference/data/cs219.lib"


def _fixture():
    d = np.load(os.path.join(GOLD, "cs219_probs.npz"))
    q = np.load(os.path.join(GOLD, "query_hhm_prepared.npz"))
    return d["lib"], d["q_profile"], d["pav"], np.ascontiguousarray(q["qp"][:-1])


def make_db(prof, n_db, seed):
    """Column-state database around a byte profile (220, Lq): a third of the sequences follow the best state of a
    query window with mutation rates 0.15..0.95 and a few indels, the rest is random."""
    rng = np.random.default_rng(seed)
    Lq = prof.shape[1]
    best = prof[:219].argmax(axis=0)
    lens = rng.integers(20, 500, n_db)
    offs = np.zeros(n_db + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    seqs = rng.integers(0, 219, offs[-1]).astype(np.uint8)
    for n in range(0, n_db, 3):
        mut = rng.uniform(0.15, 0.95)
        q, t, L = int(rng.integers(0, Lq // 2)), int(rng.integers(0, 10)), lens[n]
        while t < L and q < Lq:
            r = rng.random()
            if r < 0.02:
                q += int(rng.integers(1, 5))
            elif r < 0.04:
                t += int(rng.integers(1, 5))
            else:
                if rng.random() > mut:
                    seqs[offs[n] + t] = best[q]
                t += 1
                q += 1
    seqs[rng.integers(0, len(seqs), 50)] = 219   # the ANY state occurs too
    return seqs, offs, lens.astype(np.int32)


def ref_prefilter_db(ref, qp, pav, seqs, offs, **kw):
    from pyhhv import capi
    par = dict(capi.PREFILTER_DEFAULTS)
    par.update(kw)
    n_db = len(offs) - 1
    out = np.zeros(n_db, dtype=np.int32)
    f = ref.lib.ref_prefilter_db
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                  C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    f.restype = C.c_int
    m = f(qp.ctypes.data, pav.ctypes.data, qp.shape[0], seqs.ctypes.data, offs.ctypes.data, n_db, 4, par["gap_open"],
          par["gap_extend"], par["score_offset"], par["bit_factor"], par["evalue_thresh"], par["evalue_coarse_thresh"],
          par["smax_thresh"], par["min_hits"], par["maxnumdb"], out.ctypes.data, n_db)
    return out[:m]


def test_flog2_fpow2_bitwise(ref):
    from pyhhv import capi
    r = capi.load_runner()
    ref.lib.ref_flog2.argtypes = [C.c_float]
    ref.lib.ref_flog2.restype = C.c_float
    ref.lib.ref_fpow2.argtypes = [C.c_float]
    ref.lib.ref_fpow2.restype = C.c_float
    rng = np.random.default_rng(5)
    xs = np.concatenate([np.exp(rng.uniform(-30, 12, 4000)), np.arange(1, 600), [0.0, -1.0, 1.0, 2.0 ** -126]]).astype(np.float32)
    for x in xs:
        assert np.float32(r.hhvr_flog2(x)).tobytes() == np.float32(ref.lib.ref_flog2(x)).tobytes(), x
    for x in np.concatenate([rng.uniform(-140, 140, 4000), np.arange(-60, 3)]).astype(np.float32):
        assert np.float32(r.hhvr_fpow2(x)).tobytes() == np.float32(ref.lib.ref_fpow2(x)).tobytes(), x


@pytest.mark.skipif(not os.path.exists(REF_CS219), reason="reference data not present")
def test_context_library_parser_matches_reference():
    from pyhhv import capi
    lib, _, _, _ = _fixture()
    assert np.array_equal(capi.read_context_library(REF_CS219), lib)


def test_query_profile_matches_golden():
    from pyhhv import capi
    lib, q_profile, pav, qp = _fixture()
    got = capi.prefilter_profile(qp, pav, lib, 50, 4)
    assert got.shape == q_profile.shape and np.array_equal(got, q_profile)
    assert np.all(got[219] == 49)


@pytest.mark.parametrize("seed", range(3))
def test_query_profile_matches_reference_live(ref, seed):
    from pyhhv import capi, synth
    lib = _fixture()[0]
    Lq = [17, 64, 250][seed]
    qp = np.ascontiguousarray(synth.make_query(900 + seed, Lq)[0][:-1])
    pav = synth.PB.copy()
    want = np.zeros((220, Lq), dtype=np.uint8)
    ref.lib.ref_prefilter_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    ref.lib.ref_prefilter_profile(qp.ctypes.data, pav.ctypes.data, Lq, 50 - seed, 4 - seed % 2, want.ctypes.data)
    assert np.array_equal(capi.prefilter_profile(qp, pav, lib, 50 - seed, 4 - seed % 2), want)


SELECT_CASES = [dict(), dict(min_hits=5), dict(min_hits=5, maxnumdb=7), dict(min_hits=30, smax_thresh=40),
                dict(min_hits=3, evalue_thresh=1e-3, evalue_coarse_thresh=10.0), dict(min_hits=2000)]


@pytest.mark.parametrize("case", range(len(SELECT_CASES)))
def test_selection_steps_match_reference_prefilter_db(oracle, ref, case):
    """SelectFirst / SelectSecond on the oracle's kernel scores == the reference's whole prefilter_db."""
    from pyhhv import capi
    kw = SELECT_CASES[case]
    lib, prof, pav, qp = _fixture()
    Lq = qp.shape[0]
    seqs, offs, lens = make_db(prof, 900, 40 + case)
    want = ref_prefilter_db(ref, qp, pav, seqs, offs, **kw)
    n = len(lens)
    ung = np.array([oracle.lib.hho_ungapped_score(u8(prof), Lq, u8(seqs[offs[k]:offs[k + 1]].copy()), int(lens[k]), 50)
                    for k in range(n)], dtype=np.int32)
    subset = capi.prefilter_select_first(ung, lens, Lq, **kw)
    sw = np.array([oracle.lib.hho_sw_score(u8(prof), Lq, u8(seqs[offs[k]:offs[k + 1]].copy()), int(lens[k]), 24, 4, 50, 32)
                   for k in subset], dtype=np.int32)
    ids, ev = capi.prefilter_select_second(sw, subset, lens, Lq, **kw)
    assert len(want) > 0 and np.array_equal(ids, want)
    assert len(ev) == len(ids)


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(SELECT_CASES)))
def test_gpu_first_stage_on_device_equals_host_selection(case):
    """hhv_prefilter_first (scores, length correction, sort and cut on the device) returns the ids hhv::Prefilter::SelectFirst
    derives from the same scores on the host - same set, same order."""
    from pyhhv import capi
    kw = {k: v for k, v in SELECT_CASES[case].items() if k in ("min_hits", "smax_thresh")}
    lib, prof, pav, qp = _fixture()
    Lq = qp.shape[0]
    seqs, offs, lens = make_db(prof, 5000, 300 + case)
    c = capi.Context()
    db = c.prefilter_upload_db(seqs, offs)
    ung = c.prefilter_scores(db, prof, OFFSET, gapped=False)
    want = capi.prefilter_select_first(ung, lens, Lq, **kw)
    r = capi.load_runner()
    got = c.prefilter_first(db, prof, OFFSET, r.hhvr_flog2(float(Lq)), 4, kw.get("smax_thresh", 10), kw.get("min_hits", 100))
    assert np.array_equal(got, want) and len(got) >= min(kw.get("min_hits", 100), 5000)
    c.prefilter_free_db(db)
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(SELECT_CASES)))
def test_gpu_prefilter_db_matches_reference(ref, case):
    from pyhhv import capi
    kw = SELECT_CASES[case]
    lib, prof, pav, qp = _fixture()
    seqs, offs, lens = make_db(prof, 3000, 70 + case)
    want = ref_prefilter_db(ref, qp, pav, seqs, offs, **kw)
    c = capi.Context()
    ids, ev, passed1 = capi.prefilter_db(c, seqs, offs, lib, qp, pav, **kw)
    c.close()
    assert np.array_equal(ids, want)
    assert passed1 >= len(ids) and np.all(ev >= 0)
