#!/usr/bin/env python3
"""Prefilter fixture: the 219 context-state probabilities the reference parses from its own
data/cs219.lib (cs::ContextLibrary + TransformToLin, via oracle/_ref/libhhref.so::ref_cs219_probs), and the byte
profile Prefilter::stripe_query_profile builds from the real data/query.hhm profile (tests/golden/query_hhm_prepared.npz).

Writes tests/golden/cs219_probs.npz {lib (219,20) float64, q_profile (220,431) uint8, pav (20,) float32}.
Build-container only (needs oracle/_ref/libhhref.so built from /root/reference).
    python tests/golden/make_prefilter_fixture.py
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libhhref.so"))
    lib.ref_cs219_probs_f64.argtypes = [C.c_void_p]
    probs = np.zeros((219, 20), dtype=np.float64)
    assert lib.ref_cs219_probs_f64(probs.ctypes.data) == 219
    d = np.load(os.path.join(ROOT, "tests", "golden", "query_hhm_prepared.npz"))
    qp = np.ascontiguousarray(d["qp"][:-1])          # rows p[0..L-1], the rows stripe_query_profile reads
    Lq = qp.shape[0]
    pav = qp[1:].mean(axis=0).astype(np.float32)
    pav /= pav.sum()
    out = np.zeros((220, Lq), dtype=np.uint8)
    lib.ref_prefilter_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.ref_prefilter_profile(qp.ctypes.data, pav.ctypes.data, Lq, 50, 4, out.ctypes.data)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cs219_probs.npz"), lib=probs, q_profile=out, pav=pav)
    print("lib", probs.shape, "profile", out.shape, "mean byte", out.mean())


if __name__ == "__main__":
    main()
