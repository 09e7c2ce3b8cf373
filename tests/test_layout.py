"""Packed record layout: product packer (C, hh-suite_amd/csrc/hhv_pack.cpp) == numpy mirror, and
the fast_log2 tables the product uploads == the oracle's (which are pinned to the reference)."""
import numpy as np
import pytest

from pyhhv import capi, pack, synth


def test_pack_template_matches_numpy_mirror():
    for seed, L in ((1, 1), (2, 7), (3, 64), (4, 301)):
        p, tr = synth.make_template(seed, L)
        a = capi.pack_profile(p, tr, index=seed)
        rec, off = pack.pack_stream([p], [tr])
        m = rec.view(np.int32).copy()
        m[0, 0] = seed
        assert np.array_equal(a.view(np.int32), m[:L + 1])
        meta = a.view(np.int32)[:, 27]
        assert meta[0] == -2 ** 31 and meta[L] == (L | 0x40000000)
        assert np.array_equal(meta[1:L], np.arange(1, L, dtype=np.int32))


def test_pack_query_matches_numpy_mirror():
    for seed, L in ((5, 3), (6, 64), (7, 65), (8, 300), (9, 431)):
        p, tr = synth.make_query(seed, L)
        a = capi.pack_profile(p, tr, index=-1)
        q = pack.pack_query(p, tr)
        R, P = pack.strips_for(L)
        assert q.shape[0] == 64 * R * P and 64 * R * P >= L and R <= 5
        assert np.array_equal(a.view(np.int32), q[:L].view(np.int32))
        assert not q[L:].any()


def test_record_operand_slots():
    """Record j holds exactly the operands of a DP cell (src/hhviterbialgorithm.cpp:222-228)."""
    p, tr = synth.make_template(11, 9)
    rec = capi.pack_profile(p, tr, index=-1)
    for j in range(1, 10):
        r = rec[j - 1]
        assert np.array_equal(r[:20], p[j])
        assert r[20] == tr[j - 1, 0] and r[21] == tr[j - 1, 2] and r[22] == tr[j - 1, 5]
        assert r[23] == tr[j - 1, 6] and r[24] == tr[j - 1, 3]
        assert r[25] == tr[j, 4] and r[26] == tr[j, 1]


def test_fast_log2_tables_match_oracle(oracle):
    lg2, diff = capi.fast_log2_tables()
    for b in range(1024):
        x = np.uint32(0x3F800000 | (b << 13)).view(np.float32)
        assert np.float32(oracle.fast_log2(float(x))) == lg2[b]
        x1 = np.uint32(0x3F800000 | (b << 13) | 1).view(np.float32)
        want = np.float32(np.float32(0.0) + lg2[b]) + np.float32(diff[b] * np.float32(1.0))
        assert np.float32(oracle.fast_log2(float(x1))) == np.float32(want)


def test_bt_layout_roundtrip():
    rng = np.random.default_rng(0)
    Lq, Lt, R = 70, 33, 2
    mask = (rng.random((Lq + 1, Lt + 1)) < 0.2).astype(np.uint8)
    mask[0, :] = 0
    mask[:, 0] = 0
    buf = np.zeros(((Lt + 1 + 1), 64, 8), dtype=np.uint8)
    pack.matrix_to_bt(mask, 0, R, buf)
    assert np.array_equal(pack.celloff_bits(buf, 0, Lq, Lt, R), mask)


def test_packed_db_file_matches_stream(tmp_path):
    """hhv_db_write: file = header + L table + exactly the record stream of the numpy mirror."""
    tps, ttrs = zip(*[synth.make_template(50 + k, 5 + 7 * k) for k in range(6)])
    path = str(tmp_path / "db.hhvpdb")
    Ls = capi.db_write(path, tps, ttrs)
    raw = open(path, "rb").read()
    assert raw[:8] == b"HHVPDB01"
    n, recdw = np.frombuffer(raw[8:16], dtype=np.int32)
    nrec = int(np.frombuffer(raw[16:24], dtype=np.int64)[0])
    assert (n, recdw) == (6, 28) and nrec == int((Ls + 1).sum()) + 1
    assert np.array_equal(np.frombuffer(raw[64:64 + 4 * n], dtype=np.int32), Ls)
    body = np.frombuffer(raw[64 + 4 * n:], dtype=np.int32).reshape(nrec, 28)
    rec, off = pack.pack_stream(list(tps), list(ttrs))
    assert np.array_equal(body, rec.view(np.int32))


def test_packer_refuses_negative_profile_values_and_stores_plus_zero():
    """The kernel's log2f4 shifts the exponent of a column product out without masking the sign (one v_alignbit_b32,
    viterbi_lane.h): profile values must be >= 0.  The packer - every host path into the engine goes through it - reports
    negative values and turns -0.0f into +0.0f (x + 0.0: products, sums and log2f4 of the reference come out the same)."""
    from pyhhv import capi, synth
    p, tr = synth.make_template(77, 9)
    rec = capi.pack_profile(p, tr, index=3)
    assert rec.shape == (10, 28)
    bad = p.copy()
    bad[4, 7] = -1e-30
    with pytest.raises(capi.HhvError, match="negative profile value"):
        capi.pack_profile(bad, tr, index=3)
    with pytest.raises(capi.HhvError, match="negative profile value"):
        capi.pack_profile(bad, tr)
    z = p.copy()
    z[2, :] = -0.0
    z[5, 3] = -0.0
    rec = capi.pack_profile(z, tr, index=0)
    assert not np.any(np.signbit(rec[1:, :20])) and np.all(rec[2, :20] == 0.0)
    keep = np.ones_like(z, dtype=bool)
    keep[2, :] = False
    keep[5, 3] = False
    assert np.array_equal(rec[1:, :20][keep[1:]], p[1:][keep[1:]])
