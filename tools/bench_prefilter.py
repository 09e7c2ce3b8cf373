"""Throughput of the prefilter kernels on a synthetic column-state database.
usage: python tools/bench_prefilter.py [n_db] [Lq] [check_n]   -> one JSON line (cells = Lq * db residues)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "hh-suite_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
from pyhhv import capi  # noqa: E402


def run(n_db=1000000, Lq=300, check_n=300):
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
    return out


def main():
    n_db = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
    Lq = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    check_n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    print(json.dumps(run(n_db, Lq, check_n)))


if __name__ == "__main__":
    main()
