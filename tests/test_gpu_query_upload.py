"""hhv_set_query leaves the packed query in its pinned staging block and the next hhv_align_async moves it to the device with one
upload kernel (which also sets the stream kernel's ticket counter and reports to the host through a mapped word when the staging
block may be written again).  What must hold: the alignment sees the LAST query set; a loop of queries of changing length that never
waits for the device gives the results of one query at a time; hhv_hits behind an alignment uses the profile that alignment used;
HHV_QUERY_COPY=1 (the copy operation of the earlier rounds) gives the same.  Reference analogue: HMMSimd::MapOneHMM(q) before every
search, src/hhhmmsimd.cpp:73-79, src/hhblits.cpp:1136."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _set(n=300, seed=7100):
    from pyhhv import synth
    rng = np.random.default_rng(seed)
    tps, ttrs = [], []
    for k in range(n):
        p, tr = synth.make_template(seed + k, int(rng.integers(30, 260)))
        tps.append(p)
        ttrs.append(tr)
    return tps, ttrs


def _direct(queries, tps, ttrs, local):
    """every query in a context of its own, synchronously"""
    from pyhhv import capi
    out = []
    for q, qtr in queries:
        c = capi.Context(local=local)
        c.set_query(q, qtr)
        ts = c.upload(tps, ttrs)
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        out.append((np.array(res["score"]), np.array(res["i2"]), np.array(res["j2"]), hits.tobytes()))
        ts.free()
        c.close()
    return out


@pytest.mark.parametrize("local", [0, 1])
def test_last_query_set_is_the_one_aligned_and_loops_do_not_mix_queries(local):
    from pyhhv import capi, synth
    tps, ttrs = _set()
    queries = [synth.make_query(910 + k, L) for k, L in enumerate((120, 300, 45, 700, 300, 64))]   # one strip, short arrays, three strips
    want = _direct(queries, tps, ttrs, local)
    c = capi.Context(local=local)
    ts = c.upload(tps, ttrs)
    # two queries set, none aligned in between: the second is the one on the device
    c.set_query(*queries[0])
    c.set_query(*queries[1])
    res = c.align(ts, backtrace=True)
    assert np.array_equal(res["score"], want[1][0]) and np.array_equal(res["i2"], want[1][1])
    assert c.hits(ts).tobytes() == want[1][3]
    # a loop that never waits: set_query waits only for the previous upload kernel; the last step's results are the last query's
    for rep in range(3):
        for k, q in enumerate(queries):
            c.set_query(*q)
            c.align_async(ts, backtrace=True)
            c.hits(ts, fetch=False)
        c.sync()
        k = len(queries) - 1
        assert c.hits(ts).tobytes() == want[k][3]
    # every query once more, checked one by one (device block grown and shrunk in between)
    for k in (3, 2, 5, 0, 4, 1):
        c.set_query(*queries[k])
        res = c.align(ts, backtrace=True)
        assert np.array_equal(res["score"], want[k][0]), k
        assert np.array_equal(res["j2"], want[k][2]), k
        assert c.hits(ts).tobytes() == want[k][3], k
    if os.environ.get("HHV_QUERY_COPY") == "1":   # (the copy operation overwrites the device block at once: the rest is the upload kernel's)
        ts.free()
        c.close()
        return
    # the next query is set while the last alignment's hits have not been asked for: they are made with the profile of the alignment
    c.set_query(*queries[4])
    c.align_async(ts, backtrace=True)
    c.set_query(*queries[1])          # (same length: hhv_hits refuses a query of another length) the block is untouched until the next alignment
    assert c.hits(ts).tobytes() == want[4][3]
    res = c.align(ts, backtrace=True)
    assert np.array_equal(res["score"], want[1][0])
    c.set_query(*queries[0])
    with pytest.raises(capi.HhvError):
        c.hits(ts)                    # another length: no backtrace of the current query
    ts.free()
    c.close()


def test_copy_operation_switch_gives_the_same_results():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HHV_QUERY_COPY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                        "tests/test_gpu_query_upload.py::test_last_query_set_is_the_one_aligned_and_loops_do_not_mix_queries"],
                       cwd=os.path.dirname(here), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
