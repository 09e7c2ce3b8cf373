#!/usr/bin/env python3
"""tools/mfma_emission_study.py -- VERDICT r1 item 8 (exploratory): could the 20-term emission product run on the matrix pipe?

v_mfma_f32_32x32x2_f32 is an exact-f32 FMA chain (one rounding per term); the reference multiplies and adds separately in
four partial sums (src/hhviterbi.h:126-161).  Before writing a kernel for it, this script measures on the CPU what the other
rounding does to the RESULTS: the oracle's DP is run twice on the same pairs - emission as the reference computes it, and
as an fmaf chain (hho_set_emission_mode) - and compared: score differences, end points, alignment start, number of steps,
whole paths (checksums), Hit scores.  The acceptance rule of the verdict: indices exact on the soak, |dscore| <= 1e-4.
usage: python tools/mfma_emission_study.py [n_pairs] [Lq]      (CPU only)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "hh-suite_amd"), os.path.join(ROOT, "oracle")]
import pyoracle  # noqa: E402
from pyhhv import synth  # noqa: E402


def run(n=4000, Lq=300, local=0, homolog_every=3, seed=1):
    o = pyoracle.Oracle()
    o.lib.hho_set_emission_mode.argtypes = [C.c_int]
    par = pyoracle.make_params(local=local)
    rng = np.random.default_rng(seed)
    qf, qtr = synth.make_query(7000 + seed, Lq)
    tps, ttrs = [], []
    for k in range(n):
        L = int(rng.integers(40, 400))
        p, tr = synth.make_homolog(8000 + k, qf, L=L, mut=0.1 + 0.8 * rng.random()) if k % homolog_every == 0 else synth.make_template(8000 + k, L)
        tps.append(p)
        ttrs.append(tr)
    thr = os.cpu_count() or 1
    o.lib.hho_set_emission_mode(0)
    a = o.bench_hits(par, qf, qtr, tps, ttrs, threads=thr)
    o.lib.hho_set_emission_mode(1)
    b = o.bench_hits(par, qf, qtr, tps, ttrs, threads=thr)
    o.lib.hho_set_emission_mode(0)
    d = np.abs(a["score"].astype(np.float64) - b["score"].astype(np.float64))
    dh = np.abs(a["hit_score"].astype(np.float64) - b["hit_score"].astype(np.float64))
    ends = (a["i2"] != b["i2"]) | (a["j2"] != b["j2"])
    starts = (a["i1"] != b["i1"]) | (a["j1"] != b["j1"])
    paths = a["path_hash"] != b["path_hash"]
    order_a = np.lexsort((np.arange(n), -a["hit_score"].astype(np.float64)))[:500]
    order_b = np.lexsort((np.arange(n), -b["hit_score"].astype(np.float64)))[:500]
    return {"pairs": n, "Lq": Lq, "mode": "local" if local else "global",
            "scores_bitwise_different": int(np.sum(a["score"] != b["score"])), "max_abs_score_diff": float(d.max()),
            "mean_abs_score_diff": float(d.mean()), "max_abs_hit_score_diff": float(dh.max()),
            "end_points_different": int(ends.sum()), "alignment_starts_different": int(starts.sum()),
            "paths_different": int(paths.sum()), "top500_sets_differ_by": int(len(set(order_a.tolist()) ^ set(order_b.tolist())) // 2),
            "top500_order_identical": bool(np.array_equal(order_a, order_b)),
            "significant_hits_with_different_path": int(np.sum(paths & (a["hit_score"] > 20.0)))}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    Lq = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    out = [run(n, Lq, local=0), run(n, Lq, local=1, seed=2)]
    print(json.dumps(out, indent=1))
