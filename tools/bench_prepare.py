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


HBM_PEAK, HBM_ACHIEVABLE = 8.0e12, 6.3e12


def reference_rate(base, q_pav, pb, threads, reps):
    """PrepareTemplateHMM of the reference (oracle/ref_hmm_harness.cpp ref_prepare_raw_timed, oracle/_ref) on the distinct raw
    templates, `threads` OpenMP threads: templates per second."""
    import ctypes as C
    import pyoracle as po
    if not po.have_ref():
        return None
    lib = po.Ref().lib
    f = lib.ref_prepare_raw_timed
    f.restype = C.c_double
    f.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    n = len(base)
    L = base[0][1].shape[0] - 1
    ff = np.ascontiguousarray(np.stack([b[0] for b in base]), np.float32)
    tt = np.ascontiguousarray(np.stack([b[1] for b in base]), np.float32)
    nn = np.ascontiguousarray(np.stack([b[2] for b in base]), np.float32)
    nh = np.ascontiguousarray([b[3] for b in base], np.float32)
    qp = np.ascontiguousarray(q_pav, np.float32)
    gap = np.ascontiguousarray(po.DEFAULT_GAP, np.float32)
    pc = np.ascontiguousarray(po.DEFAULT_PC, np.float32)
    pbv = np.ascontiguousarray(pb, np.float32)
    chk = C.c_double(0)
    t = f(n, reps, L, ff.ctypes.data, tt.ctypes.data, nn.ctypes.data, nh.ctypes.data, qp.ctypes.data, gap.ctypes.data, pc.ctypes.data, 1,
          pbv.ctypes.data, threads, C.addressof(chk))
    return {"threads": threads, "templates": n * reps, "seconds": t, "templates_per_s": n * reps / t}


def run(n=100000, ref_sample=0, ref_threads=0):
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
    kms = []
    t0 = time.perf_counter()
    for _ in range(reps):
        ts = c.prepare(raw, Ls, par, q_pav, ts=ts)
        kms.append(c.last_kernel_ms())
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
    k_ms = float(np.mean(kms)) if kms and all(v > 0 for v in kms) else t_prep * 1e3
    # spot check against the oracle
    k = 3
    p, tro, pv = po.oracle_prepare(o, 1, *base[idx[k]], pb, R, q_pav=q_pav)
    ok = bool(np.array_equal(c.records_of(ts, k).view(np.uint32),
                             capi.pack_profile(np.ascontiguousarray(p[:-1]), tro, index=k).view(np.uint32)))
    out = {"templates": n, "Lt": Lt, "Lq": Lq, "upload_raw_s": t_upload, "prepare_ms": t_prep * 1e3, "prepare_kernel_ms": k_ms,
           "bytes": prep_bytes, "bytes_note": "algorithmic: 128 B raw column read + 112 B record written per column (profile and transitions stay in LDS)",
           "prepare_GBps": prep_bytes / (k_ms * 1e-3) / 1e9, "frac_of_hbm_peak": prep_bytes / (k_ms * 1e-3) / HBM_PEAK,
           "frac_of_hbm_achievable": prep_bytes / (k_ms * 1e-3) / HBM_ACHIEVABLE, "prepare_templates_per_s": n / (k_ms * 1e-3),
           "align_ms": t_align * 1e3, "prepare_plus_align_cells_per_s": n * Lq * Lt / (t_prep + t_align),
           "records_match_oracle": ok}
    c.rawset_free(raw)
    c.close()
    if ref_sample and ref_threads:
        try:
            for th, key in ((1, "reference_1core"), (ref_threads, "reference")):
                r = reference_rate(base[:ref_sample], q_pav, pb, th, max(1, 4 * th // 4))
                if r:
                    out[key] = r
            if "reference" in out:
                out["reference_templates_per_s"] = out["reference"]["templates_per_s"]
                out["reference_cores"] = out["reference"]["threads"]
                out["reference_kind"] = "reference (oracle/_ref: HMM::AddTransitionPseudocounts .. IncludeNullModelInHMM, PrepareTemplateHMM's sequence)"
        except Exception as e:  # noqa: BLE001
            out["reference_error"] = repr(e)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    ref_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    print(json.dumps(run(n, 256 if ref_threads else 0, ref_threads)))


if __name__ == "__main__":
    main()
