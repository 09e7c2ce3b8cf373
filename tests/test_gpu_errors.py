"""Error behaviour of the C ABI on a GPU box: every misuse returns a negative status with a message (never exits, never
crashes) and leaves the context usable - the reference's fatal paths (exit(3) in hhhmmsimd.cpp:88-91, hhviterbi.cpp:140-144)
become return codes."""
import numpy as np
import pytest

from pyhhv import synth

pytestmark = pytest.mark.gpu


def small_set(c, n=3, Lq=40):
    qp, qtr = synth.make_query(1, Lq)
    tps, ttrs = zip(*[synth.make_template(10 + k, 20 + k) for k in range(n)])
    return qp, qtr, list(tps), list(ttrs)


def test_state_and_argument_errors():
    from pyhhv import capi
    c = capi.Context(local=1)
    qp, qtr, tps, ttrs = small_set(c)
    ts = c.upload(tps, ttrs)
    with pytest.raises(capi.HhvError, match="query"):      # align before hhv_set_query
        c.align(ts)
    c.set_query(qp, qtr)
    res = c.align(ts)                                        # the context survived
    assert len(res) == 3
    with pytest.raises(capi.HhvError):                       # hits before a backtrace run
        c.hits(ts)
    c.align(ts, backtrace=True)
    hits = c.hits(ts)
    lib = capi.load()
    ns = np.zeros(1, np.int32)
    buf = np.zeros(2, np.int32)
    assert int(hits["nsteps"][0]) + 1 > 2
    rc = lib.hhv_hit_path(c.h, ts.h, 0, 2, buf.ctypes.data, buf.ctypes.data, None, None, ns.ctypes.data_as(capi.c_int_p))
    assert rc == 0 and ns[0] == hits["nsteps"][0] and buf[1] > 0   # cap too small: truncated copy, full length reported
    rc = lib.hhv_hit_path(c.h, ts.h, 99, 2, buf.ctypes.data, buf.ctypes.data, None, None, ns.ctypes.data_as(capi.c_int_p))
    assert rc < 0                                            # template index out of range
    c2 = capi.Context(local=1)
    c2.set_query(qp, qtr)
    with pytest.raises(capi.HhvError):                       # template set of another context
        c2.align(ts)
    c2.close()
    assert len(c.align(ts)) == 3
    ts.free()
    c.close()


def test_limits_are_reported_not_fatal(tmp_path):
    from pyhhv import capi
    c = capi.Context(local=1)
    qp, qtr, tps, ttrs = small_set(c)
    c.set_query(qp, qtr)
    # prepare: amino-acid pseudocount mode the device code does not restate
    z = np.load(__file__.replace("test_gpu_errors.py", "golden/gonnet_pb_R.npz"))
    raws = [synth.make_raw_hmm(5, 30)]
    raw, Ls = c.upload_raw([r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
    par = capi.prep_params(z["pb"], z["R"], pc=(3, 1.0, 1.5, 1.0))
    with pytest.raises(capi.HhvError, match="pcm"):
        c.prepare(raw, Ls, par, synth.PB)
    c.rawset_free(raw)
    # MAC realignment: a template beyond the LDS row state is no limit any more (row state in global memory)
    Lt = 2100
    tp = np.zeros((Lt + 1, 20), np.float32)
    tl = np.zeros((Lt + 1, 7), np.float32)
    ms = c.mac_realign(qp, capi.linear_transitions(qtr, True), [tp], [tl], None)
    assert len(ms.hits) == 1
    ms.free()
    # prefilter: state > 219 in the database, subset id out of range
    with pytest.raises(capi.HhvError, match="219"):
        c.prefilter_upload_db(np.array([1, 2, 250], np.uint8), np.array([0, 3], np.int64))
    db = c.prefilter_upload_db(np.array([1, 2, 3, 4], np.uint8), np.array([0, 2, 4], np.int64))
    prof = np.full((220, 10), 40, np.uint8)
    with pytest.raises(capi.HhvError, match="subset"):
        c.prefilter_scores(db, prof, 50, subset=np.array([0, 2], np.int32))
    assert np.array_equal(c.prefilter_scores(db, prof, 50), np.array([0, 0], np.int32))
    c.prefilter_free_db(db)
    # database files: garbage is rejected
    bad = tmp_path / "bad.db"
    bad.write_bytes(b"not a database" * 10)
    lib = capi.load()
    import ctypes as C
    h = C.c_void_p()
    assert lib.hhv_db_open(c.h, str(bad).encode(), C.byref(h)) < 0 and b"not a packed" in lib.hhv_last_error()
    with pytest.raises(capi.HhvError, match="raw template database"):
        c.rawdb_open(bad)
    assert len(c.align(c.upload(tps, ttrs))) == 3             # still alive
    c.close()
