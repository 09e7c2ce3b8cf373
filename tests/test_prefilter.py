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
