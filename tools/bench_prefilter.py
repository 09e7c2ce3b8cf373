"""Throughput of the prefilter kernels on a synthetic column-state database.
usage: python tools/bench_prefilter.py [n_db] [Lq] [check_n] [ref_threads]   -> one JSON line (cells = Lq * db residues);
ref_threads > 0: the reference's AVX2 kernels with its OpenMP loop on that many host threads beside it"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hh-suite_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from pyhhv import capi  # noqa: E402


def reference_rates(prof, Lq, seqs, offs, threads, sw_every=10):
    """The reference's AVX2 kernels (Prefilter::ungapped_sse_score / swStripedByte, src/hhprefilter.cpp:214-353) over the same
    database with the reference's OpenMP loop (oracle/ref_prefilter_harness.cpp ref_prefilter_scores_timed, oracle/_ref),
    `threads` threads; gapless over every sequence, Smith-Waterman over every sw_every-th (as the GPU entry)."""
    import pyoracle
    if not pyoracle.have_ref():
        return None
    lib = pyoracle.Ref().lib
    f = lib.ref_prefilter_scores_timed
    f.restype = C.c_double
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    prof = np.ascontiguousarray(prof)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    out = {"threads": threads, "vector_bytes": int(lib.ref_prefilter_vecbytes())}
    chk = C.c_long(0)
    n = len(offs) - 1
    t = min(f(prof.ctypes.data, Lq, seqs.ctypes.data, offs.ctypes.data, n, 50, 24, 4, 0, threads, C.addressof(chk)) for _ in range(2))
    out["gapless_cells_per_s"] = Lq * float(offs[-1]) / t
    out["gapless_s"] = t
    # every sw_every-th sequence, compacted (the harness walks a contiguous offset table)
    ids = np.arange(0, n, sw_every)
    lens = (offs[1:] - offs[:-1])[ids]
    so = np.zeros(len(ids) + 1, dtype=np.int64)
    so[1:] = np.cumsum(lens)
    ss = np.concatenate([seqs[offs[i]:offs[i + 1]] for i in ids])
    t = min(f(prof.ctypes.data, Lq, ss.ctypes.data, so.ctypes.data, len(ids), 50, 24, 4, 1, threads, C.addressof(chk)) for _ in range(2))
    out["sw_cells_per_s"] = Lq * float(so[-1]) / t
    out["sw_s"] = t
    return out


def run(n_db=1000000, Lq=300, check_n=300, ref_threads=0):
    rng = np.random.default_rng(7)
    lens = np.clip(rng.gamma(2.2, 140.0, n_db), 30, 2000).astype(np.int64)
    offs = np.zeros(n_db + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    seqs = rng.integers(0, 219, offs[-1], dtype=np.uint8)
    prof = np.clip(rng.normal(42, 9, (220, Lq)), 0, 255).astype(np.uint8)
    c = capi.Context()
    db = c.prefilter_upload_db(seqs, offs)
    out = {"n_db": n_db, "Lq": Lq, "residues": int(offs[-1])}
    for name, gapped, subset in (("ungapped", False, None), ("gapped", True, np.arange(0, n_db, 10, dtype=np.int32))):
        c.prefilter_scores(db, prof, 50, gapped=gapped, subset=subset)
        t0 = time.perf_counter()
        sc = c.prefilter_scores(db, prof, 50, gapped=gapped, subset=subset)
        wall = time.perf_counter() - t0
        kms = c.last_kernel_ms()
        res = int(lens.sum() if subset is None else lens[subset].sum())
        out[name] = {"kernel_ms": round(kms, 3), "wall_ms": round(wall * 1e3, 3), "cells_per_s": Lq * res / (kms * 1e-3),
                     "score_mean": float(sc.mean()), "score_max": int(sc.max())}
        if check_n:
            import pyoracle
            orc = pyoracle.Oracle()
            u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
            ids = (np.arange(check_n) * 7919) % len(sc)
            bad = 0
            for k in ids:
                sid = k if subset is None else subset[k]
                s = np.ascontiguousarray(seqs[offs[sid]:offs[sid + 1]])
                ref = (orc.lib.hho_sw_score(u8(prof), Lq, u8(s), len(s), 24, 4, 50, 32) if gapped else
                       orc.lib.hho_ungapped_score(u8(prof), Lq, u8(s), len(s), 50))
                bad += int(ref != sc[k])
            out[name]["checked"] = check_n
            out[name]["mismatches"] = bad
    c.prefilter_free_db(db)
    c.close()
    if ref_threads:
        try:
            out["reference"] = reference_rates(prof, Lq, seqs, offs, ref_threads)
        except Exception as e:  # noqa: BLE001
            out["reference"] = {"error": repr(e)}
    return out


def main():
    n_db = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    Lq = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    check_n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    ref_threads = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    print(json.dumps(run(n_db, Lq, check_n, ref_threads)))


if __name__ == "__main__":
    main()
