#!/usr/bin/env python3
"""One hhblits-like search iteration, every stage on the GPU, database resident in HBM (SURVEY.md 8f rows chained):

    prefilter over the cs219 database (N3)  ->  prepare the survivors from the resident raw HMMs (N2)
    ->  Viterbi + backtrace + Hit scores (hot path)  ->  top hits  ->  MAC realignment (N4)

usage: python tools/bench_pipeline.py [n_db] [survivors] [realign]      -> one JSON line with per-stage wall times.
Synthetic database: 512 distinct raw HMMs / column-state sequences replicated (the work is the same as for distinct ones)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyhhv import capi, synth  # noqa: E402


def run(n_db=200000, survivors=10000, n_realign=500):
    Lq, Lt, distinct = 300, 300, 512
    z = np.load(os.path.join(ROOT, "tests", "golden", "gonnet_pb_R.npz"))
    pb, R = z["pb"], z["R"]
    lib = np.load(os.path.join(ROOT, "tests", "golden", "cs219_probs.npz"))["lib"]
    import pyoracle as po
    orc = po.Oracle()
    fq, trq, nq, nhq = synth.make_raw_hmm(7, Lq)
    q_p, q_tr, q_pav = po.oracle_prepare(orc, 0, fq, trq, nq, nhq, pb, R)     # query preparation stays on the host
    qp = np.ascontiguousarray(q_p[:-1])
    rng = np.random.default_rng(1)

    # ---- database (built once): raw HMMs; every 16th one is related to the query
    base = []
    for k in range(distinct):
        f, tr, neff, nh = synth.make_raw_hmm(1000 + k, Lt)
        if k % 16 == 0:
            mix = 0.8 * fq[1:Lt + 1].astype(np.float64) + 0.2 * f[1:Lt + 1].astype(np.float64)
            f[1:Lt + 1] = (mix / mix.sum(axis=1, keepdims=True)).astype(np.float32)
        base.append((f, tr, neff, nh))
    idx = np.arange(n_db) % distinct
    prof = capi.prefilter_profile(np.ascontiguousarray(qp[:-1]), q_pav, lib)
    best = prof[:219].argmax(axis=0)
    base_seq = []
    for k in range(distinct):
        s = rng.integers(0, 219, Lt).astype(np.uint8)
        if k % 16 == 0:
            keep = rng.random(Lt) < 0.7
            s[keep] = best[:Lt][keep]
        base_seq.append(s)
    seqs = np.concatenate([base_seq[i] for i in idx])
    offs = np.arange(n_db + 1, dtype=np.int64) * Lt

    c = capi.Context(local=1, shift=-0.03, corr=0.1, ss_mode=0)
    t0 = time.perf_counter()
    raw, Ls_all = c.upload_raw([base[i][0] for i in idx], [base[i][1] for i in idx], [base[i][2] for i in idx],
                               [base[i][3] for i in idx])
    pfdb = c.prefilter_upload_db(seqs, offs)
    t_load = time.perf_counter() - t0
    par = capi.prep_params(pb, R)
    q_lin = capi.linear_transitions(q_tr, True)
    t_lin_base = [capi.linear_transitions(po.oracle_prepare(orc, 1, *base[k], pb, R, q_pav=q_pav)[1], False) for k in range(distinct)]

    flog2_Lq = capi.load_runner().hhvr_flog2(float(Lq))

    def search():
        t, dev = {}, {}
        t0 = time.perf_counter()
        c.set_query(qp, q_tr)
        prof = capi.prefilter_profile(np.ascontiguousarray(qp[:-1]), q_pav, lib)
        # first stage entirely on the device (scores, length correction, sort, cut): only the surviving ids come back
        sub = c.prefilter_first(pfdb, prof, 50, flog2_Lq, 4, smax_thresh=1000, min_hits=survivors)   # forces `survivors` to pass
        t["prefilter_gapless_ms"] = (time.perf_counter() - t0) * 1e3
        dev["prefilter_gapless"] = c.last_kernel_ms()
        t1 = time.perf_counter()
        sw = c.prefilter_scores(pfdb, prof, 50, gapped=True, gap_init=24, gap_extend=4, subset=sub)
        dev["prefilter_sw"] = c.last_kernel_ms()
        ids, ev = capi.prefilter_select_second(sw, sub, Ls_all, Lq, min_hits=survivors, maxnumdb=survivors)
        t["prefilter_sw_select_ms"] = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter()
        ts = c.prepare_subset(raw, Ls_all, par, q_pav, ids)
        t["prepare_subset_ms"] = (time.perf_counter() - t1) * 1e3
        dev["prepare_subset"] = c.last_kernel_ms()
        t1 = time.perf_counter()
        c.align(ts, backtrace=True)
        dev["viterbi_dp"] = c.last_kernel_ms()
        hits = c.hits(ts)
        t["viterbi_backtrace_hits_ms"] = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter()
        top = np.argsort(-hits["score"], kind="stable")[:n_realign]
        # Viterbi alignments stay on the device; entry = position in the compact list of realigned templates
        mac_in = [(e, 1, 0, 0, 0, 0, -1, None, None, int(pos)) for e, pos in enumerate(top)]
        t["select_top_ms"] = (time.perf_counter() - t1) * 1e3
        t1 = time.perf_counter()
        t_lins = [t_lin_base[idx[ids[p]]] for p in top]                  # host: powf of the realigned templates' transitions
        sc, re, *_ = capi.runner_mac_realign(c, qp, q_lin, None, t_lins, mac_in, resident=ts)
        t["mac_realign_ms"] = (time.perf_counter() - t1) * 1e3
        dev["mac_kernels"] = c.last_kernel_ms()
        t["device"] = {k: round(float(v), 3) for k, v in dev.items()}
        t["total_ms"] = (time.perf_counter() - t0) * 1e3
        t["survivors"], t["realigned"] = int(len(ids)), int(len(mac_in))
        t["related_in_top"] = int(sum(1 for p in top if idx[ids[p]] % 16 == 0))
        t["mean_mac_cols"] = float(sc[:, 5].mean())
        ts.free()
        return t

    search()                                    # warm-up (allocations, first launches)
    runs = [search() for _ in range(3)]
    best = min(runs, key=lambda r: r["total_ms"])
    out = {"n_db": n_db, "Lq": Lq, "Lt": Lt, "db_load_s": round(t_load, 2),
           "stages_ms": {k: round(v, 2) for k, v in best.items() if k.endswith("_ms")},
           "kernels_ms": dict(best["device"], note="hhv_last_kernel_ms after the stage's call: the device's share of the wall time above "
                              "(the rest: Python marshalling of this harness, the host's selection arithmetic, copies)"),
           "survivors": best["survivors"], "realigned": best["realigned"], "related_in_top": best["related_in_top"],
           "mean_mac_cols": best["mean_mac_cols"]}
    c.prefilter_free_db(pfdb)
    c.rawset_free(raw)
    c.close()
    return out


def main():
    n_db = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    survivors = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    n_realign = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    print(json.dumps(run(n_db, survivors, n_realign)))


if __name__ == "__main__":
    main()
