#!/usr/bin/env python3
"""Measurement of the N2 path (on-device PrepareTemplateHMM): raw database resident in HBM, per query
prepare + align.  Prints one JSON line: prepare time, its HBM bytes and GB/s, align time, end-to-end
templates/s.  (Auxiliary measurement; bench.py remains the contract benchmark.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyhhv import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    Lt, Lq, distinct = 300, 300, 512
    z = np.load(os.path.join(ROOT, "tests", "golden", "gonnet_pb_R.npz"))
    pb, R = z["pb"], z["R"]
    import pyoracle as po
    o = po.Oracle()
    fq, trq, nq, nhq = synth.make_raw_hmm(7, Lq)
    q_p, q_tr, q_pav = po.oracle_prepare(o, 0, fq, trq, nq, nhq, pb, R)   # query preparation stays on the host
    base = [synth.make_raw_hmm(1000 + k, Lt) for k in range(distinct)]
    idx = np.arange(n) % distinct
    c = capi.Context(local=0)
    c.set_query(q_p[:-1], q_tr)
    t0 = time.perf_counter()
    raw, Ls = c.upload_raw([base[i][0] for i in idx], [base[i][1] for i in idx], [base[i][2] for i in idx],
                           [base[i][3] for i in idx])
    t_upload = time.perf_counter() - t0
    par = capi.prep_params(pb, R)
    ts = c.prepare(raw, Ls, par, q_pav)          # first call allocates
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        ts = c.prepare(raw, Ls, par, q_pav, ts=ts)
    t_prep = (time.perf_counter() - t0) / reps
    c.align_async(ts)
    c.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        c.align_async(ts)
    c.sync()
    t_align = (time.perf_counter() - t0) / reps
    cols = n * (Lt + 1)
    # bytes moved by the fused prepare kernel: raw column 128 B read, packed record 112 B written (the mixed profile and
    # the prepared transitions stay in LDS)
    prep_bytes = cols * (128 + 112)
    # spot check against the oracle
    k = 3
    p, tro, pv = po.oracle_prepare(o, 1, *base[idx[k]], pb, R, q_pav=q_pav)
    ok = bool(np.array_equal(c.records_of(ts, k).view(np.uint32),
                             capi.pack_profile(np.ascontiguousarray(p[:-1]), tro, index=k).view(np.uint32)))
    print(json.dumps({"templates": n, "Lt": Lt, "Lq": Lq, "upload_raw_s": t_upload, "prepare_ms": t_prep * 1e3,
                      "prepare_GBps": prep_bytes / t_prep / 1e9, "prepare_templates_per_s": n / t_prep,
                      "align_ms": t_align * 1e3, "prepare_plus_align_cells_per_s": n * Lq * Lt / (t_prep + t_align),
                      "records_match_oracle": ok}))


if __name__ == "__main__":
    main()
