"""Smith-Waterman prefilter on the sequences a real search sends through it - the survivors of the gapless stage, i.e. homologs - against
random sequences: GPU kernel time (hhv_pf_sw_kernel) and the reference's AVX2 swStripedByte on the host's threads for the same sets.
The lazy-F loop of the striped algorithm (src/hhprefilter.cpp:176-203) runs longer the more similar a sequence is to the query.
usage: python tools/bench_sw_homologs.py [n_seq] [ref_threads] [Lq = Lt]  -> JSON"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pyhhv import capi, synth  # noqa: E402


def run(n=20000, threads=16, L=300, check=200):
    Lq = Lt = L
    z = np.load(os.path.join(ROOT, "tests", "golden", "gonnet_pb_R.npz"))
    pb, R = z["pb"], z["R"]
    lib = np.load(os.path.join(ROOT, "tests", "golden", "cs219_probs.npz"))["lib"]
    import pyoracle as po
    orc = po.Oracle()
    fq, trq, nq, nhq = synth.make_raw_hmm(7, Lq)
    q_p, q_tr, q_pav = po.oracle_prepare(orc, 0, fq, trq, nq, nhq, pb, R)
    qp = np.ascontiguousarray(q_p[:-1])
    prof = capi.prefilter_profile(np.ascontiguousarray(qp[:-1]), q_pav, lib)
    best = prof[:219].argmax(axis=0)
    rng = np.random.default_rng(3)
    out = {"n_seq": n, "Lq": Lq, "Lt": Lt}
    c = capi.Context()
    from bench_prefilter import reference_rates  # noqa: F401  (loads the reference library the same way)
    for name, keep_frac in (("random", 0.0), ("homologs_30pct", 0.3), ("homologs_70pct", 0.7)):
        seqs = rng.integers(0, 219, n * Lt).astype(np.uint8).reshape(n, Lt)
        if keep_frac:
            keep = rng.random((n, Lt)) < keep_frac
            seqs[keep] = np.broadcast_to(best[:Lt], (n, Lt))[keep]
        seqs = np.ascontiguousarray(seqs.reshape(-1))
        offs = np.arange(n + 1, dtype=np.int64) * Lt
        db = c.prefilter_upload_db(seqs, offs)
        sub = np.arange(n, dtype=np.int32)
        c.prefilter_scores(db, prof, 50, gapped=True, gap_init=24, gap_extend=4, subset=sub)
        sc = c.prefilter_scores(db, prof, 50, gapped=True, gap_init=24, gap_extend=4, subset=sub)
        kms = c.last_kernel_ms()
        e = {"gpu_kernel_ms": round(kms, 3), "gpu_cells_per_s": Lq * n * Lt / (kms * 1e-3), "score_mean": float(sc.mean()), "score_max": int(sc.max())}
        if check:   # a sample against the oracle (pinned to Prefilter::swStripedByte)
            u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
            p8 = np.ascontiguousarray(prof)
            bad = 0
            for q in (np.arange(check) * 97) % n:
                sq = np.ascontiguousarray(seqs[q * Lt:(q + 1) * Lt])
                bad += int(orc.lib.hho_sw_score(u8(p8), Lq, u8(sq), Lt, 24, 4, 50, 32) != sc[q])
            e["checked"], e["mismatches_vs_oracle"] = check, bad
        if po.have_ref() and threads:
            rl = po.Ref().lib
            f = rl.ref_prefilter_scores_timed
            f.restype = C.c_double
            f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
            chk = C.c_long(0)
            p = np.ascontiguousarray(prof)
            t = min(f(p.ctypes.data, Lq, seqs.ctypes.data, offs.ctypes.data, n, 50, 24, 4, 1, threads, C.addressof(chk)) for _ in range(3))
            e["reference_ms_%d_threads" % threads] = round(t * 1e3, 3)
            e["gpu_over_reference"] = t * 1e3 / kms
        out[name] = e
        c.prefilter_free_db(db)
    c.close()
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))
