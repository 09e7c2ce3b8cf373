#!/usr/bin/env python3
"""Real-profile fixture: the reference's own data/query.hhm (L=431), read and prepared by the reference's
own code (oracle/_ref/libhhref_hmm.so: HMM::Read + the PrepareQueryHMM / PrepareTemplateHMM call
sequence, -nocontxt), aligned by the reference's Viterbi (oracle/_ref/libhhref.so).

Writes  tests/golden/query_hhm_prepared.npz   prepared query / template tensors (inputs)
        tests/golden/query_hhm_golden.json    reference outputs for: the self alignment (2 strips of the
                                              multi-pass kernel) and three template windows, local + global
Build-container only (needs /root/reference).   python tests/golden/make_real_fixture.py
"""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pyoracle import Ref, make_params  # noqa: E402
from real_fixture import WINDOWS, window  # noqa: E402

HHM = "/root/reference/data/query.hhm"


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhhref.so"))
    maxres = 2000
    qp = np.zeros((maxres, 20), np.float32)
    qtr = np.zeros((maxres, 7), np.float32)
    tp = np.zeros((maxres, 20), np.float32)
    ttr = np.zeros((maxres, 7), np.float32)
    Lq, Lt = C.c_int(), C.c_int()

    def fp(a):
        return a.ctypes.data_as(C.POINTER(C.c_float))

    rc = lib.ref_prepare_hhm(HHM.encode(), HHM.encode(), maxres, fp(qp), fp(qtr), fp(tp), fp(ttr), C.byref(Lq),
                             C.byref(Lt))
    assert rc == 0, rc
    Lq, Lt = Lq.value, Lt.value
    qp, qtr, tp, ttr = qp[:Lq + 1].copy(), qtr[:Lq + 1].copy(), tp[:Lt + 1].copy(), ttr[:Lt + 1].copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "query_hhm_prepared.npz"), qp=qp, qtr=qtr, tp=tp, ttr=ttr)
    ref = Ref()
    out = {"source": "data/query.hhm of soedinglab/hh-suite v3.3.0, prepared with -nocontxt defaults", "Lq": Lq,
           "results": []}
    for local in (1, 0):
        par = make_params(local=local)
        for (a, b) in WINDOWS:
            p, tr = window(tp, ttr, a, b)
            o = ref.align_batch(par, qp, qtr, [p], [tr], replicate=True, want_path=True)[0]
            ns = o.nsteps
            out["results"].append({
                "local": local, "window": [a, b], "score_bits": int(np.float32(o.score).view(np.uint32)),
                "i2": o.i2, "j2": o.j2,
                "bt_sha256": hashlib.sha256(np.ascontiguousarray(o.bt[1:, 1:]).tobytes()).hexdigest(),
                "nsteps": ns, "matched_cols": o.matched_cols,
                "path_sha256": hashlib.sha256(np.ascontiguousarray(o.i_steps[1:ns + 1]).tobytes() +
                                              np.ascontiguousarray(o.j_steps[1:ns + 1]).tobytes() +
                                              np.ascontiguousarray(o.states[1:ns + 1]).tobytes()).hexdigest(),
                "S_sha256": hashlib.sha256(np.ascontiguousarray(o.S[1:ns + 1]).tobytes()).hexdigest(),
                "hit_score_bits": int(np.float32(o.hit_score).view(np.uint32)),
                "hit_score": float(o.hit_score)})
            print(local, (a, b), float(o.score), o.i2, o.j2, ns, float(o.hit_score))
    with open(os.path.join(ROOT, "tests", "golden", "query_hhm_golden.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))


if __name__ == "__main__":
    main()
