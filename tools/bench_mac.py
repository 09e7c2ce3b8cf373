"""MAC realignment throughput: n hits of one query realigned on the GPU (hhv::PosteriorDecoderRunner)
next to the reference's PosteriorDecoder::realign timed on one host core (oracle/_ref) on a sample of the same hits.
usage: python tools/bench_mac.py [n_hits] [Lq] [Lt] [ref_sample]   -> one JSON line
Lt = 0: template lengths of a real search, log-normal around 250 columns, 40 .. 1800 (all length classes of the launch)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyhhv import capi, synth  # noqa: E402


def reference_rate(qp, qtr, tps, ttrs, hits, threads, local=1, shift=-0.03, mact=0.3501, corr=0.1):
    """The reference's PosteriorDecoder on the given hits with PosteriorDecoderRunner's OpenMP loop over the templates
    (oracle/ref_mac_harness.cpp ref_mac_realign_timed, oracle/_ref): seconds of the loop on `threads` host threads."""
    import ctypes as C
    from pyoracle import Ref
    lib = Ref().lib
    f = lib.ref_mac_realign_timed
    f.restype = C.c_double
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                  C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    n = len(hits)
    Lts = np.array([tps[h[0]].shape[0] - 1 for h in hits], np.int32)
    col_off = np.zeros(n + 1, np.int64)
    col_off[1:] = np.cumsum(Lts + 1)
    tp = np.ascontiguousarray(np.concatenate([tps[h[0]] for h in hits]), np.float32)
    ttr = np.ascontiguousarray(np.concatenate([ttrs[h[0]] for h in hits]), np.float32)
    ends = np.ascontiguousarray([[h[2], h[3], h[4], h[5]] for h in hits], np.int32)
    ns = np.array([h[6] for h in hits], np.int32)
    path_off = np.zeros(n + 1, np.int64)
    path_off[1:] = np.cumsum(ns + 1)
    vi = np.ascontiguousarray(np.concatenate([np.asarray(h[7][:h[6] + 1], np.int32) for h in hits]))
    vj = np.ascontiguousarray(np.concatenate([np.asarray(h[8][:h[6] + 1], np.int32) for h in hits]))
    qp32, qtr32 = np.ascontiguousarray(qp, np.float32), np.ascontiguousarray(qtr, np.float32)
    chk = C.c_long(0)
    t = f(qp32.ctypes.data, qtr32.ctypes.data, qp32.shape[0] - 1, n, col_off.ctypes.data, Lts.ctypes.data, tp.ctypes.data, ttr.ctypes.data,
          int(local), shift, mact, corr, ends.ctypes.data, ns.ctypes.data, path_off.ctypes.data, vi.ctypes.data, vj.ctypes.data,
          int(threads), C.addressof(chk))
    return t, int(chk.value)


def run(n=500, Lq=300, Lt=300, sample=16, ref_threads=0):
    qp, qtr = synth.make_query(11, Lq)
    tps, ttrs = [], []
    lens = [Lt] * n
    if Lt <= 0:
        lens = np.clip(np.random.default_rng(3).lognormal(np.log(250.0), 0.7, n), 40, 1800).astype(int).tolist()
    for k in range(n):
        tp, ttr = synth.make_homolog(100 + k, qp, L=lens[k], mut=0.2 + 0.6 * (k % 7) / 7.0)
        tps.append(tp)
        ttrs.append(ttr)
    c = capi.Context(local=1, shift=-0.03, corr=0.1)
    c.set_query(qp, qtr)
    ts = c.upload(tps, ttrs)
    c.align(ts, backtrace=True)
    vh = c.hits(ts)
    hits = []
    for k in range(n):
        _, i_s, j_s, st, S = c.hit_path(ts, k)
        ns = int(vh["nsteps"][k])
        hits.append((k, 1, int(vh["i1"][k]), int(vh["j1"][k]), int(vh["i2"][k]), int(vh["j2"][k]), ns, i_s, j_s))
    q_lin = capi.linear_transitions(qtr, True)
    t_lins = [capi.linear_transitions(t, False) for t in ttrs]
    capi.runner_mac_realign(c, qp, q_lin, tps[:4], t_lins[:4], hits[:4])          # warm-up
    t0 = time.perf_counter()
    sc, re, o_i, o_j, o_s, o_S, o_P = capi.runner_mac_realign(c, qp, q_lin, tps, t_lins, hits)
    wall = time.perf_counter() - t0
    kms = c.last_kernel_ms()
    import ctypes
    t3 = (ctypes.c_double * 3)()
    capi.load_runner().hhvr_mac_last_timing(t3)
    # the same hits with the profiles AND the Viterbi alignments taken from the resident template set the Viterbi stage just
    # searched (hhv_mac_realign_tset, inputs without a path): only the linear transitions cross PCIe
    res_hits = [(k, 1, h[2], h[3], h[4], h[5], -1, None, None) for k, h in enumerate(hits)]
    capi.runner_mac_realign(c, qp, q_lin, None, t_lins[:4], res_hits[:4], resident=ts)
    t0 = time.perf_counter()
    r_sc, r_re, r_i, r_j, r_s, r_S, r_P = capi.runner_mac_realign(c, qp, q_lin, None, t_lins, res_hits, resident=ts)
    wall_res = time.perf_counter() - t0
    kms_res = c.last_kernel_ms()
    t3r = (ctypes.c_double * 3)()
    capi.load_runner().hhvr_mac_last_timing(t3r)
    same = bool(np.array_equal(sc, r_sc) and re.tobytes() == r_re.tobytes() and np.array_equal(o_i, r_i) and o_P.tobytes() == r_P.tobytes())
    cells = float(Lq) * float(sum(lens))
    out = {"n_hits": n, "Lq": Lq, "Lt": Lt if Lt > 0 else "lognormal(250, 0.7) in 40..1800, max %d" % max(lens), "gpu_kernels_ms": round(kms, 3), "gpu_wall_ms_incl_host_masks": round(wall * 1e3, 2),
           "host_ms_masks_realign_fetch": [round(v, 2) for v in t3],
           "resident_set": {"gpu_kernels_ms": round(kms_res, 3), "gpu_wall_ms": round(wall_res * 1e3, 2),
                            "runner_ms_inputs_realign_fetch": [round(v, 2) for v in t3r], "runner_ms": round(sum(t3r), 2),
                            "identical_to_staged": same},
           "gpu_hits_per_s": n / (kms * 1e-3), "gpu_cells_per_s": cells / (kms * 1e-3),
           "mean_nsteps": float(sc[:, 0].mean()), "mean_sum_of_probs": float(re[:, 1].mean())}
    if sample:
        from pyoracle import Ref, ref_mac_realign

        class V:
            pass
        ref = Ref()
        bad = 0
        t_ref = 0.0
        for k in range(0, n, max(1, n // sample)):
            v = V()
            v.nsteps, v.i2, v.j2, v.i_steps, v.j_steps = hits[k][6], hits[k][4], hits[k][5], hits[k][7], hits[k][8]
            t1 = time.perf_counter()
            r = ref_mac_realign(ref, qp, qtr, tps[k], ttrs[k], v, local=1)
            t_ref += time.perf_counter() - t1
            ok = (tuple(sc[k]) == (r.nsteps, r.i1, r.j1, r.i2, r.j2, r.matched_cols) and
                  np.float64(re[k, 0]).tobytes() == np.float64(r.Pforward).tobytes() and
                  o_P[k, 1:r.nsteps + 1].tobytes() == r.P[1:r.nsteps + 1].tobytes() and
                  np.array_equal(o_i[k, 1:r.nsteps + 1], r.i_steps[1:r.nsteps + 1]))
            bad += int(not ok)
            out.setdefault("checked", 0)
            out["checked"] += 1
        out["mismatches_vs_reference"] = bad
        out["ref_cpu_ms_per_hit_1core"] = round(t_ref / out["checked"] * 1e3, 3)
        out["ref_cpu_hits_per_s_1core"] = out["checked"] / t_ref
    if ref_threads:
        try:
            # the whole batch on `ref_threads` host threads, the way the reference parallelises it (one template per thread at a time)
            t_ref, chk = reference_rate(qp, qtr, tps, ttrs, hits, ref_threads)
            out["reference_openmp"] = {"threads": ref_threads, "hits": n, "seconds": t_ref, "hits_per_s": n / t_ref, "ms": t_ref * 1e3,
                                       "sum_nsteps": chk, "gpu_sum_nsteps": int(sc[:, 0].sum())}
        except Exception as e:  # noqa: BLE001
            out["reference_openmp"] = {"error": repr(e)}
    c.close()
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    Lq = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    Lt = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    sample = int(sys.argv[4]) if len(sys.argv) > 4 else 16
    ref_threads = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    print(json.dumps(run(n, Lq, Lt, sample, ref_threads)))


if __name__ == "__main__":
    main()
