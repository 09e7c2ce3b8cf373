#!/usr/bin/env python3
"""tools/wave_times.py -- who finishes when: per-workgroup entry / exit times of hhv_stream_kernel (measurement build
`-DHHV_EXP_WAVETIME`, HHV_LIB must point at it; hh-suite_amd/csrc/hhv_stream_kernel.h hhv_dbg_wave).

All workgroups of a launch are resident at once, so the launch lasts as long as its slowest workgroup.  With one fixed range
of the template stream per workgroup (rounds 1-3, today's -DHHV_NO_QUEUE build) this showed exits 11.3 .. 14.2 ms in the
headline launch - what led to the work queue (NOTES_r3.md 7); with the queue it shows how close together the workgroups end.
Per workload: the spread of the record counts, of the durations and of the time per record, by XCD, per SIMD pair, and how the
exit times relate to the record counts.
    make lib_variant NAME=wt FLAGS=-DHHV_EXP_WAVETIME       (add -DHHV_NO_QUEUE for the fixed ranges)
    HHV_LIB=.../libhhviterbi_wt.so python tools/wave_times.py [fixed|fixed10000|zipf_local|zipf_sorted_global ...]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))


def probe(lib, n_waves):
    buf = np.zeros(4 * n_waves, dtype=np.uint64)
    rc = lib.hhv_debug_wave(buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.c_int(n_waves))
    if rc != 0:
        raise SystemExit("hhv_debug_wave failed: %d" % rc)
    return buf.reshape(n_waves, 4)


def describe(name, w):
    start, end, M, ids = (w[:, 0].astype(np.float64), w[:, 1].astype(np.float64), w[:, 2].astype(np.float64), w[:, 3])
    live = M > 0
    start, end, M, ids = start[live], end[live], M[live], ids[live]
    t0 = start.min()
    dur = (end - start) / 100.0          # us (100 MHz)
    fin = (end - t0) / 100.0
    per = dur / M * 1e3                  # ns per record
    hw = (ids & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = ((ids >> np.uint64(32)) & np.uint64(0xF)).astype(np.int64)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    se = (hw >> 13) & 7
    out = {"workload": name, "workgroups": int(live.sum()),
           "records": {"mean": float(M.mean()), "max_over_mean": float(M.max() / M.mean()), "min_over_mean": float(M.min() / M.mean())},
           "launch_us": float(fin.max()), "start_spread_us": float((start.max() - t0) / 100.0),
           "duration_us": {"mean": float(dur.mean()), "min": float(dur.min()), "max": float(dur.max()), "p01": float(np.percentile(dur, 1)), "p99": float(np.percentile(dur, 99))},
           "exit_us": {"first": float(fin.min()), "p10": float(np.percentile(fin, 10)), "median": float(np.median(fin)), "p90": float(np.percentile(fin, 90)), "last": float(fin.max())},
           "ns_per_record": {"mean": float(per.mean()), "min": float(per.min()), "max": float(per.max()), "std_over_mean": float(per.std() / per.mean())},
           "corr_duration_records": float(np.corrcoef(dur, M)[0, 1]) if M.std() > 0 else None}
    out["by_xcc_ns_per_record"] = {int(x): float(per[xcc == x].mean()) for x in np.unique(xcc)}
    out["by_xcc_last_exit_us"] = {int(x): float(fin[xcc == x].max()) for x in np.unique(xcc)}
    # the slowest 1 %: are they the largest ranges, or slow places?
    k = max(1, len(dur) // 100)
    slow = np.argsort(-fin)[:k]
    out["last_1pct"] = {"records_over_mean": float(M[slow].mean() / M.mean()), "ns_per_record_over_mean": float(per[slow].mean() / per.mean())}
    # two workgroups of one SIMD: the partner's exit vs own
    key = (xcc * 8 + se) * 64 + cu * 4 + simd
    order = np.argsort(key, kind="stable")
    ks = key[order]
    same = ks[1:] == ks[:-1]
    if same.any():
        a, b = order[:-1][same], order[1:][same]
        out["simd_pairs"] = {"pairs": int(same.sum()), "mean_abs_exit_gap_us": float(np.abs(fin[a] - fin[b]).mean())}
    # a least-squares line: duration = c0 + c1 * records (c0: what a workgroup pays regardless of its range)
    if M.std() > 0:
        c1, c0 = np.polyfit(M, dur, 1)
        out["fit_duration_us"] = {"per_record_ns": float(c1 * 1e3), "constant_us": float(c0), "ratio_to_mean_rate": float(c1 / (dur.mean() / M.mean()))}
    return out


def main():
    import torch
    from pyhhv import capi, synth, synth_stream
    lib = capi.load()
    if not hasattr(lib, "hhv_debug_wave"):
        raise SystemExit("HHV_LIB is not a -DHHV_EXP_WAVETIME build")
    lib.hhv_debug_wave.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    which = sys.argv[1:] or ["fixed", "zipf_local", "zipf_sorted_global"]
    device = torch.device("cuda:0")
    Lq = 300
    qf, qtr = synth.make_query(0x51000000, Lq)
    res = []
    for name in which:
        if name.startswith("fixed"):
            n, local = (int(name[5:]) if len(name) > 5 else 100000), 0      # fixed / fixed10000 ...
            Ls = np.full(n, 300, dtype=np.int32)
            ids = np.arange(n)
        else:
            n, local = 200000, 1 if "local" in name else 0
            Ls = synth.zipf_lengths(0x21F, n).astype(np.int32)
            ids = np.arange(n)
            if "sorted" in name:
                ids = np.argsort(-Ls, kind="stable")
        rec, off, L2 = synth_stream.gen_stream(torch, device, ids.astype(np.int64), Ls[ids], synth.PB)
        torch.cuda.synchronize()
        c = capi.Context(local=local, device=0)
        c.set_query(qf, qtr)
        ts = c.adopt_device_stream(L2, rec.data_ptr())
        for _ in range(3):
            c.align_async(ts)
        c.sync()
        ms = c.last_kernel_ms()
        w = probe(lib, 4096)
        d = describe(name, w)
        d["kernel_ms_events"] = ms
        res.append(d)
        print(json.dumps(d))
        sys.stdout.flush()
        ts.free()
        c.close()
        del rec
    out = os.environ.get("HHV_WAVE_OUT")
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
