"""On-device PrepareTemplateHMM.
CPU: the oracle's restatement (hho_prepare) is pinned bit for bit to the reference's own
     AddTransitionPseudocounts / PreparePseudocounts / AddAminoAcidPseudocounts /
     CalculateAminoAcidBackground / IncludeNullModelInHMM (oracle/ref_hmm_harness.cpp) on synthetic raw
     HMMs and on the real data/query.hhm (committed as tests/golden/query_hhm_raw.npz).
GPU: hhv_prepare_templates must emit exactly the packed records of the oracle-prepared profiles, and a
     search on device-prepared templates must equal a search on host-prepared ones."""
import os

import numpy as np
import pytest

import pyoracle as po
from pyhhv import pack, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def gonnet():
    z = np.load(os.path.join(HERE, "golden", "gonnet_pb_R.npz"))
    return z["pb"], z["R"]


def raw_query_hhm(with_pb=False):
    """data/query.hhm as HMM::Read leaves it; pb_file = the NULL line of the file, which HMM::Read writes over the
    caller's background array (src/hhhmm.cpp:536-546) and which the reference therefore uses afterwards."""
    z = np.load(os.path.join(HERE, "golden", "query_hhm_raw.npz"))
    out = (z["f"], z["tr"], z["neff"], np.float32(z["neff_hmm"]))
    return out + (z["pb_file"],) if with_pb else out


def test_fixture_matches_reference(ref):
    pb, R = gonnet()
    pb2, R2 = po.ref_substitution_matrix(ref)
    assert np.array_equal(pb, pb2) and np.array_equal(R, R2)


def test_fpow2_bitexact(oracle, ref):
    import ctypes as C
    for lib, name in ((oracle.lib, "hho_fpow2"), (ref.lib, "ref_fpow2")):
        getattr(lib, name).restype = C.c_float
        getattr(lib, name).argtypes = [C.c_float]
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-130, 130, 4000), rng.uniform(-2, 2, 2000), [0, 1, -1, 127.99, 128, -125, -125.01, -99.999]])
    for x in xs.astype(np.float32):
        a, b = oracle.lib.hho_fpow2(float(x)), ref.lib.ref_fpow2(float(x))
        assert np.float32(a).tobytes() == np.float32(b).tobytes(), x


@pytest.mark.parametrize("seed", range(24))
def test_prepare_restatement_bitexact_synthetic(oracle, ref, seed):
    pb, R = gonnet()
    L = 5 + seed * 17
    f, tr, neff, nh = synth.make_raw_hmm(100 + seed, L)
    # (seeds 12..19: pcc != 1 - src/hhhmm.cpp:1903-1909, where pow(float, float) is the float overload: powf)
    # (seeds 20..23: pcm 3, the constant-diversity pseudocounts of :1911-1919 - pca is recomputed from pcb there)
    pc = np.array(([[2, 1.0, 1.5, 1.0], [0, 1, 1.5, 1], [1, 0.4, 1.5, 1.0], [2, 0.9, 2.0, 1.0]] if seed < 12 else
                   [[2, 1.0, 1.5, 0.8], [2, 0.9, 2.0, 1.3], [2, 1.0, 1.5, 2.0], [2, 0.7, 1.0, 0.5]] if seed < 20 else
                   [[3, 1.0, 1.5, 1.0], [3, 0.3, 4.0, 0.5], [3, 1.0, 12.0, 1.6], [3, 1.0, 2.5, 0.0]])[seed % 4], np.float32)
    gap = po.DEFAULT_GAP.copy()
    if seed % 5 == 0:
        gap[0], gap[1] = 0.3, 0.8
    rq = po.ref_prepare(ref, 0, f, tr, neff, nh, gap=gap, pc=pc)
    oq = po.oracle_prepare(oracle, 0, f, tr, neff, nh, pb, R, gap=gap, pc=pc)
    f2, tr2, neff2, nh2 = synth.make_raw_hmm(900 + seed, L + 3)
    rt = po.ref_prepare(ref, 1, f2, tr2, neff2, nh2, q_pav=rq[2], gap=gap, pc=pc, columnscore=seed % 4)
    ot = po.oracle_prepare(oracle, 1, f2, tr2, neff2, nh2, pb, R, q_pav=oq[2], gap=gap, pc=pc, columnscore=seed % 4)
    for a, b in list(zip(rq, oq)) + list(zip(rt, ot)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_prepare_restatement_bitexact_real_hhm(oracle, ref):
    _, R = gonnet()
    f, tr, neff, nh, pb = raw_query_hhm(with_pb=True)
    rq = po.ref_prepare(ref, 0, f, tr, neff, nh, pb=pb)
    oq = po.oracle_prepare(oracle, 0, f, tr, neff, nh, pb, R)
    rt = po.ref_prepare(ref, 1, f, tr, neff, nh, q_pav=rq[2], pb=pb)
    ot = po.oracle_prepare(oracle, 1, f, tr, neff, nh, pb, R, q_pav=oq[2])
    for a, b in list(zip(rq, oq)) + list(zip(rt, ot)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # and the prepared tensors are the ones of the committed real-profile fixture
    z = np.load(os.path.join(HERE, "golden", "query_hhm_prepared.npz"))
    assert np.array_equal(oq[0][1:-1], z["qp"][1:]) and np.array_equal(oq[1], z["qtr"])
    assert np.array_equal(ot[0][1:-1], z["tp"][1:]) and np.array_equal(ot[1], z["ttr"])


@pytest.mark.gpu
@pytest.mark.parametrize("columnscore,pcm,pcc", [(1, 2, 1.0), (0, 2, 1.0), (2, 1, 1.0), (3, 0, 1.0), (1, 2, 0.8), (0, 2, 1.7),
                                                 (1, 3, 1.0), (2, 3, 1.6)])
def test_gpu_prepare_matches_oracle(oracle, columnscore, pcm, pcc):
    """(pcc != 1: tau of every raw column comes from the host's powf, hhv_api_prep.cpp ensure_tau - same bits as the reference)"""
    from pyhhv import capi
    pb, R = gonnet()
    rng = np.random.default_rng(columnscore * 10 + pcm)
    fq, trq, nq, nhq = raw_query_hhm()
    q_p, q_tr, q_pav = po.oracle_prepare(oracle, 0, fq, trq, nq, nhq, pb, R)
    raws = [synth.make_raw_hmm(300 + k, int(rng.integers(1, 200))) for k in range(20)] + [(fq, trq, nq, nhq)]
    pc = (pcm, 0.7 if pcm == 1 else 1.0, 1.5, pcc)
    c = capi.Context(local=1)
    c.set_query(q_p[:-1], q_tr)
    raw, Ls = c.upload_raw([r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    par = capi.prep_params(pb, R, pc=pc, columnscore=columnscore)
    ts = c.prepare(raw, Ls, par, q_pav)
    pav = c.rawset_pav(raw, len(raws))
    host_p, host_tr = [], []
    for k, (f, tr, neff, nh) in enumerate(raws):
        p, tro, pv = po.oracle_prepare(oracle, 1, f, tr, neff, nh, pb, R, q_pav=q_pav, pc=np.array(pc, np.float32),
                                       columnscore=columnscore)
        host_p.append(np.ascontiguousarray(p[:-1]))
        host_tr.append(tro)
        want = capi.pack_profile(host_p[-1], tro, index=k)
        got = c.records_of(ts, k)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), k
        assert np.array_equal(pav[k].view(np.uint32), pv.view(np.uint32))
    # search on device-prepared templates == search on host-prepared templates
    ts2 = c.upload(host_p, host_tr)
    r1, r2 = c.align(ts), c.align(ts2)
    assert np.array_equal(r1.view(np.uint8), r2.view(np.uint8))
    # refill for a second query composition: results follow the new null model
    ts = c.prepare(raw, Ls, par, np.roll(q_pav, 3), ts=ts)
    if columnscore in (1, 3):
        assert not np.array_equal(c.records_of(ts, 0)[1:, :20], capi.pack_profile(host_p[0], host_tr[0], index=0)[1:, :20])
    c.rawset_free(raw)
    ts.free()
    ts2.free()
    c.close()


@pytest.mark.gpu
def test_gpu_prepare_length_classes(oracle):
    """The three code paths of hhv_prepare_templates (fused with small / large LDS footprint, split with the intermediate
    in HBM) in one raw set: lengths on both sides of the class borders."""
    from pyhhv import capi
    pb, R = gonnet()
    fq, trq, nq, nhq = raw_query_hhm()
    q_p, q_tr, q_pav = po.oracle_prepare(oracle, 0, fq, trq, nq, nhq, pb, R)
    lengths = [1, 447, 448, 449, 1300, 1301, 1700, 60]
    raws = [synth.make_raw_hmm(4000 + k, L) for k, L in enumerate(lengths)]
    c = capi.Context(local=1)
    c.set_query(q_p[:-1], q_tr)
    raw, Ls = c.upload_raw([r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    ts = c.prepare(raw, Ls, capi.prep_params(pb, R), q_pav)
    pav = c.rawset_pav(raw, len(raws))
    for k, (f, tr, neff, nh) in enumerate(raws):
        p, tro, pv = po.oracle_prepare(oracle, 1, f, tr, neff, nh, pb, R, q_pav=q_pav)
        want = capi.pack_profile(np.ascontiguousarray(p[:-1]), tro, index=k)
        assert np.array_equal(c.records_of(ts, k).view(np.uint32), want.view(np.uint32)), (k, lengths[k])
        assert np.array_equal(pav[k].view(np.uint32), pv.view(np.uint32)), k
    c.rawset_free(raw)
    ts.free()
    c.close()


def test_rawdb_file_layout(tmp_path):
    """hhv_rawdb_write needs no device: header, lengths, Neff_HMM, 32-dword raw columns."""
    from pyhhv import capi
    raws = [synth.make_raw_hmm(70 + k, L) for k, L in enumerate([3, 40, 17])]
    path = tmp_path / "db.hhvraw"
    Ls = capi.rawdb_write(path, [r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    blob = path.read_bytes()
    assert blob[:8] == b"HHVRAW01"
    n, dw, ncols = np.frombuffer(blob, np.int32, 2, 8).tolist() + [int(np.frombuffer(blob, np.int64, 1, 16)[0])]
    assert (n, dw, ncols) == (3, 32, int((Ls + 1).sum()))
    assert np.array_equal(np.frombuffer(blob, np.int32, 3, 64), Ls)
    assert np.array_equal(np.frombuffer(blob, np.float32, 3, 64 + 12), np.array([r[3] for r in raws], np.float32))
    cols = np.frombuffer(blob, np.float32, ncols * 32, 64 + 24).reshape(ncols, 32)
    assert np.array_equal(cols[1, :20], raws[0][0][1]) and np.array_equal(cols[Ls[0] + 1 + 5, 20:27], raws[1][1][5])
    assert len(blob) == 64 + 24 + ncols * 128


@pytest.mark.gpu
def test_gpu_rawdb_roundtrip(oracle, tmp_path):
    """A raw database file opened with hhv_rawdb_open prepares to the same records as the same HMMs uploaded directly."""
    from pyhhv import capi
    pb, R = gonnet()
    fq, trq, nq, nhq = raw_query_hhm()
    q_p, q_tr, q_pav = po.oracle_prepare(oracle, 0, fq, trq, nq, nhq, pb, R)
    raws = [synth.make_raw_hmm(600 + k, L) for k, L in enumerate([12, 300, 77, 500, 1])]
    path = tmp_path / "db.hhvraw"
    capi.rawdb_write(path, [r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    c = capi.Context(local=1)
    c.set_query(q_p[:-1], q_tr)
    par = capi.prep_params(pb, R)
    raw1, L1 = c.upload_raw([r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    raw2, L2 = c.rawdb_open(path)
    assert np.array_equal(L1, L2)
    ts1, ts2 = c.prepare(raw1, L1, par, q_pav), c.prepare(raw2, L2, par, q_pav)
    for k in range(len(raws)):
        assert np.array_equal(c.records_of(ts1, k).view(np.uint32), c.records_of(ts2, k).view(np.uint32)), k
    assert np.array_equal(c.align(ts1).view(np.uint8), c.align(ts2).view(np.uint8))
    for h in (raw1, raw2):
        c.rawset_free(h)
    ts1.free()
    ts2.free()
    c.close()


@pytest.mark.gpu
def test_gpu_prepare_subset(oracle):
    """hhv_prepare_subset: any id list (order, repeats, all three length classes) of the resident raw set gives the records
    the full preparation gives for those templates - only the template index in the header differs."""
    from pyhhv import capi
    pb, R = gonnet()
    fq, trq, nq, nhq = raw_query_hhm()
    q_p, q_tr, q_pav = po.oracle_prepare(oracle, 0, fq, trq, nq, nhq, pb, R)
    lengths = [30, 500, 7, 1400, 447, 448, 90, 1]
    raws = [synth.make_raw_hmm(5000 + k, L) for k, L in enumerate(lengths)]
    c = capi.Context(local=1)
    c.set_query(q_p[:-1], q_tr)
    raw, Ls = c.upload_raw([r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    par = capi.prep_params(pb, R)
    full = c.prepare(raw, Ls, par, q_pav)
    ids = np.array([6, 3, 3, 0, 7, 5, 1, 4], np.int32)
    sub = c.prepare_subset(raw, Ls, par, q_pav, ids)
    for pos, k in enumerate(ids):
        a, b = c.records_of(sub, pos).view(np.uint32).copy(), c.records_of(full, int(k)).view(np.uint32).copy()
        assert a[0, 0] == pos and b[0, 0] == k          # header: index inside the respective set
        a[0, 0] = b[0, 0] = 0
        assert np.array_equal(a, b), (pos, k)
    ra, rb = c.align(sub), c.align(full)
    assert np.array_equal(ra["score"].view(np.uint32), rb["score"][ids].view(np.uint32)) and np.array_equal(ra["i2"], rb["i2"][ids])
    with pytest.raises(capi.HhvError):
        c.prepare_subset(raw, Ls, par, q_pav, np.array([0, 8], np.int32))
    c.rawset_free(raw)
    full.free()
    sub.free()
    c.close()
