"""The drop-in translation unit hh-suite_amd/dropin/hhviterbirunner_hip.cpp against the reference's own
src/hhviterbirunner.cpp: both define ViterbiRunner::alignment with the signature of src/hhviterbirunner.h:50-58 and
are driven by the same harness (oracle/ref_runner_harness.cpp) the way HHblits::run drives them - .hhm texts read by
the reference's HMM::Read, prepared by the reference's PrepareQueryHMM / PrepareTemplateHMM, std::vector<HHEntry*> in,
std::vector<Hit> out.  The comparison is on the Hit objects the callers consume.

oracle/_ref/libhhref_dropin.so is built where /root/reference is present and travels to the GPU box prebuilt."""
import ctypes
import os

import numpy as np
import pytest

import hhm_text

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libhhref_dropin.so")


class RRHit(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("entry", "irep", "lastrep", "L", "nsteps", "matched_cols", "i1", "j1", "i2",
                                               "j2", "ssm1", "ssm2", "n_display")] + \
               [(n, ctypes.c_float) for n in ("score", "score_ss", "score_aass", "Neff_HMM")] + [("name", ctypes.c_char * 64)]


def _lib():
    if not os.path.exists(LIB):
        pytest.skip("oracle/_ref/libhhref_dropin.so not built (needs /root/reference at build time)")
    return ctypes.CDLL(LIB)


def run(which, query, templates, names, seq_len=None, loc=1, altali=4, ssm=2, early=0, prefilter=0, dbsize=20000,
        maxres=2000, threads=1, smin=20.0, filter_thresh=0.01, egq=0.0, egt=0.0, ssw=0.11, excl="", texcl="",
        path_cap=1400, pcm=-1, columnscore=-1):
    lib = _lib()
    fn = getattr(lib, "ref_runner_run_" + which)
    n = len(templates)
    cap = n * max(1, altali)
    hits = (RRHit * cap)()
    pi = np.zeros((cap, path_cap), dtype=np.int32)
    pj = np.zeros((cap, path_cap), dtype=np.int32)
    ps = np.zeros((cap, path_cap), dtype=np.int8)
    pS = np.zeros((cap, path_cap), dtype=np.float32)
    pSS = np.zeros((cap, path_cap), dtype=np.float32)
    texts = (ctypes.c_char_p * n)(*templates)
    lens = (ctypes.c_size_t * n)(*[len(t) for t in templates])
    nm = (ctypes.c_char_p * n)(*[x.encode() for x in names])
    if seq_len is None:
        seq_len = [int(t.split(b"LENG")[1].split()[0]) for t in templates]
    sl = np.asarray(seq_len, dtype=np.int32)
    oi = np.asarray([loc, altali, ssm, early, prefilter, dbsize, maxres, threads, pcm, columnscore], dtype=np.int32)
    of = np.asarray([smin, filter_thresh, egq, egt, ssw], dtype=np.float32)
    P = ctypes.c_void_p
    secs = ctypes.c_double(0.0)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int, P, P, P, P, P, P, ctypes.c_char_p, ctypes.c_char_p,
                   ctypes.c_int, P, ctypes.c_int, P, P, P, P, P, P]
    m = fn(query, len(query), n, ctypes.cast(texts, P), ctypes.cast(lens, P), ctypes.cast(nm, P), sl.ctypes.data,
           oi.ctypes.data, of.ctypes.data, excl.encode(), texcl.encode(), cap, ctypes.cast(hits, P), path_cap,
           pi.ctypes.data, pj.ctypes.data, ps.ctypes.data, pS.ctypes.data, pSS.ctypes.data, ctypes.addressof(secs))
    assert 0 <= m <= cap, m
    run.last_alignment_seconds = secs.value
    return [hits[k] for k in range(m)], pi[:m], pj[:m], ps[:m], pS[:m], pSS[:m]


def _ss(seed, L, keys):
    """random secondary-structure records; keys: True = all three, or a tuple out of ("dssp", "pred", "conf")"""
    d = hhm_text.random_ss(seed, L)
    return d if keys is True else {k: v for k, v in d.items() if k in keys}


def make_db(seed, Lq, n, lo, hi, ss_every=0, query_ss=False, homolog_every=2, same_len_every=0, ss_longest=0, ss_keys=True):
    rng = np.random.default_rng(seed)
    qf = hhm_text.random_columns(seed * 7 + 1, Lq)
    query = hhm_text.hhm_text("query%d" % seed, qf, seed, ss=_ss(seed, Lq, query_ss) if query_ss else None)
    texts, names = [], []
    Ls = [int(rng.integers(lo, hi + 1)) for _ in range(n)]
    if same_len_every:                  # equal lengths: the order std::sort leaves them in matters
        Ls = [(lo + hi) // 2 if k % same_len_every == 0 else L for k, L in enumerate(Ls)]
    with_ss = set(np.argsort(-np.asarray(Ls), kind="stable")[:ss_longest].tolist())   # the first SIMD batches
    for k in range(n):
        L = Ls[k]
        if homolog_every and k % homolog_every == 0:
            start = int(rng.integers(0, max(1, Lq - L + 1)))
            f = hhm_text.mutate_columns(seed * 1000 + k, qf[start:start + L] if L <= Lq else
                                        np.concatenate([qf, hhm_text.random_columns(seed + k, L - Lq)]), mut=0.3)
            if k % (2 * homolog_every) == 0 and L > 60:   # a repeat: second copy of the first half -> alternative alignments
                f[L // 2:L // 2 + L // 3] = f[:L // 3]
        else:
            f = hhm_text.random_columns(seed * 1000 + k, L)
        ss = _ss(seed * 31 + k, f.shape[0], ss_keys) if ((ss_every and k % ss_every == 0) or k in with_ss) else None
        names.append("t%d_%05d" % (seed, k))
        texts.append(hhm_text.hhm_text(names[-1], f, seed * 1000 + k, ss=ss))
    return query, texts, names


def cache_stats():
    lib = _lib()
    t, c = ctypes.c_size_t(0), ctypes.c_size_t(0)
    lib.hhviterbirunner_hip_cache_stats(ctypes.byref(t), ctypes.byref(c))
    return t.value, c.value


def cache_clear():
    _lib().hhviterbirunner_hip_cache_clear()


def compare(a, b):
    ha, hb = a[0], b[0]
    assert len(ha) == len(hb), (len(ha), len(hb))
    for k, (x, y) in enumerate(zip(ha, hb)):
        for f, _ in RRHit._fields_:
            vx, vy = getattr(x, f), getattr(y, f)
            assert vx == vy, (k, f, vx, vy, x.name, x.irep)
        ns = x.nsteps
        for arr in range(1, 4):
            assert np.array_equal(a[arr][k][1:ns + 1], b[arr][k][1:ns + 1]), (k, arr, x.name, x.irep)
        # A hit without any admissible cell (everything masked in a later round) ends at (0, 0); its single path step then
        # "scores" row 0 / column 0 of the profiles, which the reference never initialises (HMM::p[0] of a scratch HMM is
        # whatever earlier templates left there, divided by the null model once more, src/hhhmm.cpp:2069-2092): undefined
        # in the reference, so S / S_ss are compared on the steps inside the matrix only.
        inside = (a[1][k][1:ns + 1] > 0) & (a[2][k][1:ns + 1] > 0)
        for arr in (4, 5):
            assert np.array_equal(a[arr][k][1:ns + 1][inside], b[arr][k][1:ns + 1][inside]), (k, arr, x.name, x.irep)
    return len(ha)


def test_reference_runner_on_hhm_texts():
    """CPU only: the harness drives the reference's ViterbiRunner on synthetic .hhm texts; homologs score, repeats
    give alternative alignments, the sort by length holds.  (Validates the test data, not the product.)"""
    q, t, names = make_db(3, 120, 12, 60, 160)
    hits, pi, pj, ps, pS, pSS = run("cpu", q, t, names, altali=3)
    first = [h for h in hits if h.irep == 1]
    assert len(first) == 12
    assert [h.L for h in first] == sorted([h.L for h in first], reverse=True)
    assert max(h.score for h in first) > 40 and min(h.score for h in first) < 20
    assert any(h.irep > 1 for h in hits)
    for k, h in enumerate(hits):
        assert h.i2 == pi[k][1] and h.j2 == pj[k][1] and h.i1 == pi[k][h.nsteps] and h.j1 == pj[k][h.nsteps]


def test_library_exports_both_runners():
    lib = _lib()
    assert lib.ref_runner_run_cpu and lib.ref_runner_run_hip


@pytest.mark.gpu
@pytest.mark.parametrize("loc,altali,threads", [(1, 4, 1), (1, 1, 3), (1, 2, 2)])
def test_dropin_equals_reference(loc, altali, threads):
    q, t, names = make_db(11 + altali, 150, 45, 40, 260, same_len_every=5)
    ref = run("cpu", q, t, names, loc=loc, altali=altali, threads=1)
    got = run("hip", q, t, names, loc=loc, altali=altali, threads=threads)
    n = compare(ref, got)
    assert n > 45 or altali == 1


@pytest.mark.gpu
def test_dropin_resident_cache():
    """The second search of the same templates takes them from the device-resident raw cache (nothing is parsed or
    prepared on the host) - also for ANOTHER query, whose composition changes the prepared profiles - and still returns
    the reference's hits; HHV_TEMPLATE_CACHE-less behaviour (host preparation) is covered by the ssm/pcm variants below."""
    cache_clear()
    q, t, names = make_db(41, 150, 60, 40, 260)
    ref = run("cpu", q, t, names, altali=3)
    got = run("hip", q, t, names, altali=3, threads=4)
    compare(ref, got)
    n_templates, n_cols = cache_stats()
    assert n_templates == 60 and n_cols == sum(int(x.split(b"LENG")[1].split()[0]) + 1 for x in t)
    empty = [b""] * len(t)                    # the texts are not needed any more: reading one would fail
    got2 = run("hip", q, empty, names, altali=3, seq_len=[int(x.split(b"LENG")[1].split()[0]) for x in t])
    compare(ref, got2)
    q2, _, _ = make_db(42, 170, 1, 50, 50)    # another query against the resident templates
    ref3 = run("cpu", q2, t, names, altali=2)
    got3 = run("hip", q2, empty, names, altali=2, seq_len=[int(x.split(b"LENG")[1].split()[0]) for x in t])
    compare(ref3, got3)
    assert cache_stats()[0] == 60
    cache_clear()
    assert cache_stats() == (0, 0)


@pytest.mark.gpu
def test_dropin_concurrent_searches_share_the_cache():
    """hhblits_omp: several queries search the same database from different threads of one process.  The cache and the
    device context are shared, device sections are serialised, reading is not - every search must still return the
    reference's hits (ctypes releases the GIL, so the four calls really overlap)."""
    import threading
    cache_clear()
    _, t, names = make_db(91, 150, 120, 40, 260)
    queries = [make_db(92 + k, 120 + 15 * k, 1, 50, 50)[0] for k in range(4)]
    refs = [run("cpu", q, t, names, altali=2) for q in queries]
    for rnd in range(2):                       # round 0: cold cache, all four read; round 1: warm
        got = [None] * 4

        def work(k):
            got[k] = run("hip", queries[k], t, names, altali=2, threads=2)
        th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for k in range(4):
            assert got[k] is not None
            compare(refs[k], got[k])
    assert cache_stats()[0] == 120
    cache_clear()


@pytest.mark.gpu
@pytest.mark.parametrize("pcm,columnscore", [(3, -1), (0, 0), (1, 2), (2, 3), (-1, 4)])
def test_dropin_preparation_variants(pcm, columnscore):
    """pseudocount modes / null models: 0..3 / 0..3 are prepared on the device, the others by the reference's host code
    (columnscore 4) - the hits must not tell the difference.  One template carries its own NULL line: the
    reference prepares it with that background (HMM::Read overwrites pb, src/hhhmm.cpp:536-546), so it takes the host path."""
    cache_clear()
    q, t, names = make_db(51 + pcm + 7 * columnscore, 140, 30, 50, 220)
    odd = t[7].split(b"NULL   ")
    t[7] = odd[0] + b"NULL   " + odd[1].replace(b"3706\t5728", b"3500\t5900", 1)
    assert t[7] != odd[0] + b"NULL   " + odd[1]
    ref = run("cpu", q, t, names, altali=2, pcm=pcm, columnscore=columnscore)
    got = run("hip", q, t, names, altali=2, pcm=pcm, columnscore=columnscore)
    compare(ref, got)
    device = (pcm in (-1, 0, 1, 2, 3)) and (columnscore in (-1, 0, 1, 2, 3))   # (pcm 3: on the device since round 4)
    assert cache_stats()[0] == (29 if device else 0)
    cache_clear()


@pytest.mark.gpu
def test_dropin_global_equal_lengths():
    """global mode on templates of ONE length (no batch-composition quirk)"""
    q, t, names = make_db(5, 100, 24, 80, 80)
    ref = run("cpu", q, t, names, loc=0, altali=2)
    got = run("hip", q, t, names, loc=0, altali=2)
    compare(ref, got)


@pytest.mark.gpu
def test_dropin_baseline_config0_shape():
    """BASELINE.json configs[0]: a 431-column query (the length of data/query.hhm: two passes of the stream kernel) against
    64 synthetic templates of 200 columns, default parameters of hhsearch (local, altali 4, ssm 2, smin 20)."""
    cache_clear()
    q, t, names = make_db(431, 431, 64, 200, 200)
    ref = run("cpu", q, t, names)
    got = run("hip", q, t, names, threads=4)
    assert compare(ref, got) >= 64
    got2 = run("hip", q, t, names, threads=4)        # warm cache
    compare(ref, got2)
    cache_clear()


@pytest.mark.gpu
def test_dropin_long_profiles():
    """a 700-column query (three passes) against templates of up to 1500 columns: every length class of the device
    preparation (LDS-resident up to 447 / 1300 columns, two-kernel path beyond) behind the drop-in"""
    cache_clear()
    q, t, names = make_db(97, 700, 18, 300, 1500, homolog_every=3)
    ref = run("cpu", q, t, names, altali=2, maxres=2000, path_cap=2300)
    got = run("hip", q, t, names, altali=2, maxres=2000, path_cap=2300, threads=3)
    compare(ref, got)
    assert max(h.L for h in ref[0]) > 1300 and min(h.L for h in ref[0]) < 447
    cache_clear()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 3])
def test_dropin_global_mixed_lengths(threads):
    """global mode on templates of MANY lengths: in the reference the shorter templates of a SIMD batch are maximised over
    their last row only (the batch's last column is padding for them); the drop-in rebuilds the batches of
    the sorted block and marks those templates (hhv_set_global_batch), in every alternative-alignment round."""
    cache_clear()
    q, t, names = make_db(95, 120, 45, 30, 200, same_len_every=4)
    ref = run("cpu", q, t, names, loc=0, altali=3)
    got = run("hip", q, t, names, loc=0, altali=3, threads=threads)
    n = compare(ref, got)
    assert n > 45
    cache_clear()


@pytest.mark.gpu
@pytest.mark.parametrize("ssm", [2, 0, 4])
def test_dropin_secondary_structure(ssm):
    """query and some templates carry ss_pred/ss_conf/ss_dssp records: the ss mode is decided per SIMD batch of the
    sorted block (src/hhviterbirunner.cpp:14-22), so templates WITH records fall into batches without SS scoring"""
    dbs = [make_db(21, 130, 40, 50, 200, ss_every=1, query_ss=True),     # every batch scores SS
           make_db(21, 130, 40, 50, 200, ss_every=3, query_ss=True),     # no batch does, although templates carry records
           make_db(22, 130, 40, 50, 200, ss_longest=20, query_ss=True),  # the first two batches do, the third is mixed
           make_db(23, 130, 24, 50, 200, ss_every=1, query_ss=False)]    # templates with records, query without
    with_ss = []
    for (qq, tt, nn) in dbs:
        cache_clear()     # dbs[0] and dbs[1] use the same names for templates that differ in their ss records
        ref = run("cpu", qq, tt, nn, ssm=ssm, altali=2)
        got = run("hip", qq, tt, nn, ssm=ssm, altali=2)
        compare(ref, got)
        with_ss.append(sum(1 for h in ref[0] if h.score_ss != 0))
    assert with_ss[0] >= 40 and with_ss[1] == 0 and 16 <= with_ss[2] < 40 and with_ss[3] == 0, with_ss


@pytest.mark.gpu
def test_dropin_excluded_regions():
    q, t, names = make_db(31, 140, 20, 60, 220)
    ref = run("cpu", q, t, names, altali=2, excl="10-30,100-120", texcl="5-25")
    got = run("hip", q, t, names, altali=2, excl="10-30,100-120", texcl="5-25")
    compare(ref, got)


@pytest.mark.gpu
def test_dropin_early_stopping_blocks():
    """hhblits mode: blocks of 2000 templates in round 0, stop when the block's sum of 1/(1+E) falls under the cutoff
    (src/hhviterbirunner.cpp:109-111,178-188,213-247).  4300 unrelated short templates after 150 related ones: the
    reference stops after the second block; the drop-in must stop at the same place and return the same hits."""
    rng = np.random.default_rng(9)
    Lq = 90
    qf = hhm_text.random_columns(77, Lq)
    q = hhm_text.hhm_text("query", qf, 1)
    texts, names, seq_len = [], [], []
    for k in range(4300):
        L = int(rng.integers(30, 70))
        f = hhm_text.mutate_columns(k, qf[:L], 0.25) if k < 150 else hhm_text.random_columns(5000 + k, L)
        names.append("e%05d" % k)
        texts.append(hhm_text.hhm_text(names[-1], f, k))
        seq_len.append(L)
    kw = dict(seq_len=seq_len, altali=2, early=1, prefilter=1, dbsize=4300, filter_thresh=0.01, maxres=400, path_cap=200)
    ref = run("cpu", q, texts, names, **kw)
    got = run("hip", q, texts, names, threads=4, **kw)
    n = compare(ref, got)
    first = sum(1 for h in ref[0] if h.irep == 1)
    assert first in (2000, 4000), first   # stopped early, not after all 4300
