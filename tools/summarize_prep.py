#!/usr/bin/env python3
"""Condense gpurun_out/prof_prep/ (tools/profile_prep.sh) into <tag>_prep_summary.{txt,json}: kernel time, HBM bytes of the
prepare kernel from the PMC counters (FETCH_SIZE x 1024 x 2 - the gfx950 correction of MI355X_MICROARCH.md for wide coalesced
reads - and WRITE_SIZE x 1024) beside the algorithmic 128 + 112 bytes per column, and its VALU issue rate.
usage: python tools/summarize_prep.py [tag] [n_templates]"""
import collections
import csv
import glob
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r6"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_prep")
DST = os.environ.get("HHV_PROFILE_OUT", os.path.join(ROOT, "gpurun_out", "profiles_out"))
os.makedirs(DST, exist_ok=True)
PEAK_VALU = 256 * 4 * 32 * 2.4e9
HBM_PEAK, HBM_ACHIEVABLE = 8.0e12, 6.3e12


def pmc(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(SRC, sub, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for (k, _), cs in per.items():
            for c, v in cs.items():
                agg[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import srchash  # noqa: E402


stats = {}
for f in glob.glob(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "hhv_prep" in r["Name"]:
            stats[r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6}
counters = collections.defaultdict(dict)
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    for k, cs in pmc(sub).items():
        counters[k].update(cs)
bench = {}
for line in reversed(open(os.path.join(SRC, "stats.txt")).read().splitlines()):
    if line.startswith("{"):
        bench = json.loads(line)
        break
cols = N * 301
alg = cols * (128 + 112)
out = {"tag": TAG, "kernel_sources_sha1": srchash.kernel_sources_sha1(srchash.PREP), "templates": N, "Lt": 300, "algorithmic_bytes": alg, "bench_prepare": bench, "kernels": {}}
lines = ["%s_prep_summary -- on-device PrepareTemplateHMM (N2), %d templates x 300 columns, 1x MI355X (tools/profile_prep.sh)" % (TAG, N),
         "algorithmic HBM bytes per launch: %d columns x (128 B raw in + 112 B record out) = %.3f GB" % (cols, alg / 1e9)]
for k, st in stats.items():
    c = counters.get(k, {})
    e = {"avg_ms": st["avg_ms"], "min_ms": st["min_ms"], "calls": st["calls"]}
    t = st["avg_ms"] * 1e-3
    e["algorithmic_GBps"] = alg / t / 1e9
    e["frac_of_hbm_peak"] = alg / t / HBM_PEAK
    e["frac_of_hbm_achievable"] = alg / t / HBM_ACHIEVABLE
    if "FETCH_SIZE" in c:
        e["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in c:
        e["valu_wave_instr"] = c["SQ_INSTS_VALU"]
        e["valu_frac_of_issue_peak"] = c["SQ_INSTS_VALU"] * 64 / t / PEAK_VALU
    for n in ("SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "SQ_WAVES", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
        if n in c:
            e[n] = c[n]
    out["kernels"][k] = e
    lines.append("  %-60s %8.3f ms (min %.3f)  %.0f GB/s algorithmic = %.2f of 8 TB/s (%.2f of the achievable 6.3)" % (k[:60], st["avg_ms"], st["min_ms"], e["algorithmic_GBps"], e["frac_of_hbm_peak"], e["frac_of_hbm_achievable"]))
    lines.append("    counters: " + ", ".join("%s %.4g" % (n, v) for n, v in e.items() if n not in ("avg_ms", "min_ms", "calls", "algorithmic_GBps", "frac_of_hbm_peak", "frac_of_hbm_achievable")))
lines.append("bench line: " + json.dumps(bench))
open(os.path.join(DST, "%s_prep_summary.json" % TAG), "w").write(json.dumps(out, indent=1))
open(os.path.join(DST, "%s_prep_summary.txt" % TAG), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
