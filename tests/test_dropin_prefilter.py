"""The drop-in translation unit hh-suite_amd/dropin/hhprefilter_hip.cpp against the reference's own src/hhprefilter.cpp:
both define class Prefilter (src/hhprefilter.h) and are driven by oracle/ref_prefilterdb_harness.cpp the way
HHblitsDatabase drives them - an FFindexDatabase of column-state sequences in, (length, name) pairs of the templates to
search out, split into new and previously searched ones."""
import ctypes as C
import os

import numpy as np
import pytest

from test_dropin_runner import _lib
from test_prefilter import SELECT_CASES, _fixture, make_db


def write_ffindex(tmp, seqs, offs, dup_every=0):
    """cs219 ffindex pair: entry n = its residues + a terminating NUL (Prefilter::init_prefilter takes length - 1, :326);
    names carry an extension that prefilter_db strips (RemoveExtension, :560).  dup_every: every such entry appears twice
    under the same name (what `doubled` filters, :562)."""
    data, index, at = bytearray(), [], 0
    names = []
    for n in range(len(offs) - 1):
        s = bytes(seqs[offs[n]:offs[n + 1]]) + b"\0"
        name = "seq%06d.a3m" % (n if not (dup_every and n % dup_every == 1) else n - 1)
        data += s
        index.append("%s\t%d\t%d\n" % (name, at, len(s)))
        names.append(name)
        at += len(s)
    fd, fi = os.path.join(tmp, "db_cs219.ffdata"), os.path.join(tmp, "db_cs219.ffindex")
    open(fd, "wb").write(bytes(data))
    open(fi, "w").write("".join(sorted(index)))   # ffindex files are sorted by name
    return fd, fi, names


def prefilter_db(which, fd, fi, qp, pav, previous=(), threads=1, reps=1, **kw):
    from pyhhv import capi
    par = dict(capi.PREFILTER_DEFAULTS)
    par.update(kw)
    fn = getattr(_lib(), "ref_prefilterdb_run_" + which)
    ipar = np.asarray([threads, par["gap_open"], par["gap_extend"], par["score_offset"], par["bit_factor"], par["smax_thresh"],
                       par["min_hits"], par["maxnumdb"]], dtype=np.int32)
    dpar = np.asarray([par["evalue_thresh"], par["evalue_coarse_thresh"]], dtype=np.float64)
    cap = 1 << 22
    new_out, old_out = C.create_string_buffer(cap), C.create_string_buffer(cap)
    new_len, old_len = np.zeros(1 << 20, dtype=np.int32), np.zeros(1 << 20, dtype=np.int32)
    prev = (C.c_char_p * max(1, len(previous)))(*[p.encode() for p in previous])
    fn.restype = C.c_int
    P = C.c_void_p
    fn.argtypes = [C.c_char_p, C.c_char_p, P, P, C.c_int, P, P, C.c_int, P, P, P, P, P, C.c_int, C.c_int, P]
    secs = C.c_double(0.0)
    r = fn(fd.encode(), fi.encode(), qp.ctypes.data, pav.ctypes.data, qp.shape[0], ipar.ctypes.data, dpar.ctypes.data,
           len(previous), C.cast(prev, P), C.cast(new_out, P), new_len.ctypes.data, C.cast(old_out, P), old_len.ctypes.data, cap,
           reps, C.addressof(secs))
    prefilter_db.last_seconds = secs.value
    assert r >= 0, r
    n_new, n_old = r & 0xFFFFF, r >> 20
    new = list(zip(new_out.value.decode().split("\n")[:-1], new_len[:n_new].tolist()))
    old = list(zip(old_out.value.decode().split("\n")[:-1], old_len[:n_old].tolist()))
    assert len(new) == n_new and len(old) == n_old
    return new, old


def test_reference_prefilter_on_ffindex(tmp_path):
    """CPU only: the reference's Prefilter on an ffindex database written by this test (validates the test data)."""
    lib, prof, pav, qp = _fixture()
    seqs, offs, lens = make_db(prof, 600, 5)
    fd, fi, names = write_ffindex(str(tmp_path), seqs, offs)
    new, old = prefilter_db("cpu", fd, fi, qp, pav, min_hits=20)
    assert len(new) >= 20 and not old
    assert all(n.endswith(".a3m") for n, _ in new) and all(l == lens[int(n[3:9])] for n, l in new)
    prev = [new[0][0][:-4], new[3][0][:-4]]
    new2, old2 = prefilter_db("cpu", fd, fi, qp, pav, previous=prev, min_hits=20)
    assert [n for n, _ in old2] == [new[0][0], new[3][0]] and len(new2) == len(new) - 2


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(SELECT_CASES)))
def test_dropin_prefilter_equals_reference(tmp_path, case):
    kw = SELECT_CASES[case]
    lib, prof, pav, qp = _fixture()
    seqs, offs, lens = make_db(prof, 3000, 170 + case)
    fd, fi, names = write_ffindex(str(tmp_path), seqs, offs, dup_every=7 if case % 2 else 0)
    ref_new, _ = prefilter_db("cpu", fd, fi, qp, pav, **kw)
    prev = [n[:-4] for n, _ in ref_new[1::3]]
    ref = prefilter_db("cpu", fd, fi, qp, pav, previous=prev, **kw)
    got = prefilter_db("hip", fd, fi, qp, pav, previous=prev, threads=4, **kw)
    assert got == ref
    assert len(ref[0]) > 0 and len(ref[1]) > 0
