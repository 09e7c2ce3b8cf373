"""C++ host layer (hh-suite_amd/host/viterbi_runner.cpp = mirror of ViterbiRunner::alignment,
/root/reference src/hhviterbirunner.cpp:75-210) against the same control flow executed with the oracle:
alternative-alignment rounds with accumulated ExcludeAlignment masks, irep / lastrep / smin logic."""
import numpy as np
import pytest

from common import same_float, workload
from pyoracle import make_params

pytestmark = pytest.mark.gpu


def region_mask(Lq, Lt, excl, texcl):
    m = np.zeros((Lq + 1, Lt + 1), dtype=np.uint8)
    for a, b in excl:
        m[max(1, a):min(b, Lq) + 1, 1:] = 1
    for a, b in texcl:
        m[1:, max(1, a):min(b, Lt) + 1] = 1
    return m


def reference_rounds(oracle, par, qf, qtr, tps, ttrs, altali, smin, excl=(), texcl=()):
    """ViterbiRunner::alignment (:104-189) with the oracle standing in for Viterbi::Align & co."""
    Lq = qf.shape[0] - 1
    n = len(tps)
    masks = [None] * n
    regions = bool(excl or texcl)
    if regions:
        masks = [region_mask(Lq, t.shape[0] - 1, excl, texcl) for t in tps]
    todo = list(range(n))
    out = []
    for r in range(altali):
        nxt = []
        for k in todo:
            a = oracle.align(par, qf, qtr, tps[k], ttrs[k], celloff=masks[k] if (r > 0 or regions) else None, want_path=True)
            out.append((k, r + 1, a))
            if float(a.hit_score) > smin:
                nxt.append(k)
                m = masks[k] if masks[k] is not None else np.zeros((Lq + 1, tps[k].shape[0]), dtype=np.uint8)
                masks[k] = oracle.exclude_alignment(Lq, tps[k].shape[0] - 1, a.i_steps, a.j_steps, a.nsteps, mask=m)
        todo = nxt
        if not todo:
            break
    return out


@pytest.mark.parametrize("local", [1, 0])
def test_runner_alt_alignments(oracle, local):
    from pyhhv import capi
    par = make_params(local=local)
    Lq = 180
    qf, qtr, tps, ttrs = workload(60 + local, Lq, 10, 90, 220, homolog_every=2)
    altali, smin = 4, 20.0
    hits, i_s, j_s, st, S = capi.runner_alignment(qf, qtr, tps, ttrs, loc=local, altali=altali, smin=smin)
    want = reference_rounds(oracle, par, qf, qtr, tps, ttrs, altali, smin)
    assert len(hits) == len(want)
    assert len(hits) > len(tps), "workload must trigger at least one alternative alignment"
    for h, (k, irep, a), ii, jj, ss, sc in zip(hits, want, i_s, j_s, st, S):
        assert (h["entry"], h["irep"]) == (k, irep)
        assert same_float(h["score"], a.hit_score)
        assert h["lastrep"] == int(float(a.hit_score) <= smin)
        ns = a.nsteps
        assert (h["i2"], h["j2"], h["nsteps"], h["matched_cols"]) == (a.i2, a.j2, ns, a.matched_cols)
        assert (h["i1"], h["j1"]) == (a.i_steps[ns], a.j_steps[ns])
        assert np.array_equal(ii[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(jj[1:ns + 1], a.j_steps[1:ns + 1])
        assert np.array_equal(ss[1:ns + 1], a.states[1:ns + 1]) and np.array_equal(sc[1:ns + 1], a.S[1:ns + 1])


def test_runner_excluded_regions(oracle):
    """-excl / -template_excl (src/hhviterbirunner.cpp:157-164,291-329): masked already in the first round."""
    from pyhhv import capi
    par = make_params(local=1)
    Lq = 150
    qf, qtr, tps, ttrs = workload(71, Lq, 6, 90, 200, homolog_every=1)
    hits, i_s, j_s, st, S = capi.runner_alignment(qf, qtr, tps, ttrs, loc=1, altali=2, smin=20.0,
                                                  exclstr="10-25,100-120", template_exclstr="40-60")
    want = reference_rounds(oracle, par, qf, qtr, tps, ttrs, 2, 20.0, excl=((10, 25), (100, 120)), texcl=((40, 60),))
    assert len(hits) == len(want)
    for h, (k, irep, a) in zip(hits, want):
        assert (h["entry"], h["irep"]) == (k, irep) and same_float(h["score"], a.hit_score)
        assert (h["i2"], h["j2"], h["nsteps"]) == (a.i2, a.j2, a.nsteps)


@pytest.mark.parametrize("local,shards", [(1, 2), (0, 3), (1, 8)])
def test_sharded_runner_equals_single_runner(local, shards):
    """hhv::ShardedViterbiRunner (template database sharded by hhv_shard_plan, one hhv_ctx and one host thread per shard,
    logical shards on this one GPU) returns exactly what one ViterbiRunner returns: same hits in the same order, all
    alternative-alignment rounds, paths included."""
    from pyhhv import capi
    Lq = 160
    qf, qtr, tps, ttrs = workload(80 + shards, Lq, 150, 30, 260, homolog_every=3)
    one = capi.runner_alignment(qf, qtr, tps, ttrs, loc=local, altali=3, smin=20.0, device=0)
    many = capi.runner_alignment(qf, qtr, tps, ttrs, loc=local, altali=3, smin=20.0, device=-shards)
    assert len(one[0]) > len(tps), "workload must trigger alternative alignments"
    assert np.array_equal(one[0], many[0])
    for a, b in zip(one[1:], many[1:]):
        assert np.array_equal(a, b)
    plan = capi.shard_plan([t.shape[0] - 1 for t in tps], shards)
    assert len(set(plan.tolist())) == shards


def test_cpp_example_runs():
    """The host classes from plain C++ (examples/search_example.cpp): builds against the two shared objects and finds
    the related templates, Viterbi and MAC."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", root, "example"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(root, "build", "search_example"), "48"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    lines = out.stdout.strip().splitlines()
    assert "realigned" in lines[0] and len(lines) >= 12
    assert all("MAC q" in ln for ln in lines[1:])
