#!/usr/bin/env python3
"""tools/bench_dropin_stages.py [n_db] [threads] -- the three stages of an hhblits iteration through the reference's OWN
class interfaces, reference translation unit against drop-in translation unit (oracle/_ref/libhhref_dropin.so), wall time
of the call alone, results checked identical:
  Prefilter::prefilter_db                    on n_db column-state sequences (default 1 000 000, mean length 260)
  PosteriorDecoderRunner::executeComputation on the hits of a 400-template search
(ViterbiRunner::alignment: tools/bench_dropin.py).  Run on the GPU box."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))
import numpy as np

import hhm_text
from test_dropin_prefilter import prefilter_db, write_ffindex
from test_dropin_realign import compare as compare_realign, realign
from test_prefilter import _fixture, make_db


def big_db(prof, n_db, seed):
    """make_db for a million sequences: the related third is built from 3000 prototypes (the python loop of make_db is slow)"""
    seqs0, offs0, lens0 = make_db(prof, 3000, seed)
    rng = np.random.default_rng(seed)
    pick = rng.integers(0, 3000, n_db)
    lens = lens0[pick].astype(np.int64)
    offs = np.zeros(n_db + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    seqs = np.empty(offs[-1], dtype=np.uint8)
    for k in range(n_db):
        s = seqs0[offs0[pick[k]]:offs0[pick[k] + 1]]
        seqs[offs[k]:offs[k + 1]] = s
    noise = rng.random(len(seqs)) < 0.1          # so that copies of a prototype do not score identically
    seqs[noise] = rng.integers(0, 219, int(noise.sum())).astype(np.uint8)
    return seqs, offs, lens.astype(np.int32)


def main():
    n_db = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    out = {"threads": threads}
    lib, prof, pav, qp = _fixture()
    seqs, offs, lens = big_db(prof, n_db, 7)
    with tempfile.TemporaryDirectory() as tmp:
        fd, fi, _ = write_ffindex(tmp, seqs, offs)
        ref = prefilter_db("cpu", fd, fi, qp, pav, threads=threads, reps=3)
        t_ref = prefilter_db.last_seconds
        got = prefilter_db("hip", fd, fi, qp, pav, threads=threads, reps=3)
        t_hip = prefilter_db.last_seconds
    same = sorted(ref[0]) == sorted(got[0])      # with several threads the reference's order of equal e-values varies
    out["prefilter_db"] = {"n_db": n_db, "residues": int(offs[-1]), "Lq": int(qp.shape[0]), "selected": len(ref[0]),
                           "reference_s": round(t_ref, 4), "dropin_s": round(t_hip, 4), "same_selection": bool(same)}

    L = 300
    qf = hhm_text.random_columns(1, L)
    q = hhm_text.hhm_text("query", qf, 1)
    texts, names = [], []
    for k in range(400):
        f = hhm_text.mutate_columns(k, qf, 0.35 + 0.001 * k)
        if k % 3 == 0:
            f[L // 2:L // 2 + L // 3] = f[:L // 3]
        names.append("m%05d" % k)
        texts.append(hhm_text.hhm_text(names[-1], f, k))
    kw = dict(altali=2, maxres=L + 100, path_cap=2 * L + 10, threads=threads)
    r_ref = realign("cpu", q, texts, names, **kw)
    t_ref = realign.last_seconds
    r_hip = realign("hip", q, texts, names, **kw)
    r_hip = realign("hip", q, texts, names, **kw)
    t_hip = realign.last_seconds
    compare_realign(r_ref, r_hip)
    out["realign"] = {"hits": len(r_ref[0]), "Lq": L, "Lt": L, "reference_s": round(t_ref, 4), "dropin_s": round(t_hip, 4),
                      "hits_identical": True}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
