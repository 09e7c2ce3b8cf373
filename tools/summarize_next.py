#!/usr/bin/env python3
"""Condense gpurun_out/prof_next/ (tools/profile_next.sh) into profiles/<tag>_next_rows_summary.{txt,json} and copy the
kernel-stats tables: rocprofv3 evidence for the widened rows (prefilter N3, MAC realignment N4), with the VALU-issue roofline
of every kernel: executed VALU wave-instructions x 64 / duration against 78.6 T lane-ops/s (256 CU x 4 SIMD x 32 lanes x 2.4 GHz,
the same peak bench.py uses for the Viterbi kernel) - and per DP cell for the prefilter kernels.
usage: python tools/summarize_next.py [tag]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r3"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_next")
DST = os.environ.get("HHV_PROFILE_OUT", os.path.join(ROOT, "profiles"))   # on the GPU box: gpurun_out/profiles_out
PEAK = 256 * 4 * 32 * 2.4e9


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {}


def pmc_per_launch(*dirs):
    """kernel name -> counter -> mean value per launch, over the given --pmc passes"""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(SRC, d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(f)):
                per_dispatch[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
            for (k, _), cs in per_dispatch.items():
                for c, v in cs.items():
                    agg[k][c].append(v)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items() if "hhv" in k}


def stats_table(sub):
    out = {}
    for f in glob.glob(os.path.join(SRC, sub, "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out[r["Name"]] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "min_ms": float(r["MinNs"]) / 1e6,
                              "total_ms": float(r["TotalDurationNs"]) / 1e6}
        shutil.copy(f, os.path.join(DST, "%s_%s_kernel_stats.csv" % (TAG, sub)))
    return out


def short(name):
    return name.split("(")[0].replace("void ", "").replace("hhv::", "")


def main():
    os.makedirs(DST, exist_ok=True)
    pf, mac, mac2k = (last_json(os.path.join(SRC, n + ".txt")) for n in ("prefilter", "mac", "mac2k"))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import srchash
    summary = {"tag": TAG, "kernel_sources_sha1": srchash.kernel_sources_sha1(srchash.NEXT_ROWS), "valu_issue_peak_lane_ops_per_s": PEAK, "bench_prefilter": pf, "bench_mac_500": mac, "bench_mac_2000": mac2k, "kernels": {}}
    txt = ["profiles/%s_next_rows_summary.txt -- rocprofv3 evidence for the widened rows (SURVEY.md 8f N3, N4), 1x MI355X" % TAG,
           "commands: tools/profile_next.sh (rocprofv3 --kernel-trace --stats; separate --pmc passes); summary by tools/summarize_next.py",
           "VALU-issue roofline: SQ_INSTS_VALU x 64 / kernel duration against %.1f T lane-ops/s" % (PEAK / 1e12), ""]
    for title, sub, pdirs, cells in (("N3 prefilter, 1e6 sequences, Lq = 300 (tools/bench_prefilter.py)", "prefilter", ("prefilter_pmc", "prefilter_pmc2"),
                                      {"ungapped": pf.get("Lq", 300) * pf.get("residues", 0), "sw": None}),
                                     ("N4 MAC realignment, 500 hits Lq = 300 x Lt = 300, local (tools/bench_mac.py)", "mac", ("mac_pmc", "mac_pmc2"), {})):
        st, pm = stats_table(sub), pmc_per_launch(*pdirs)
        txt.append(title + ":")
        for name in sorted(pm, key=lambda k: -st.get(k, {}).get("total_ms", 0.0)):
            d, s = pm[name], st.get(name, {})
            if not s:
                continue
            valu = d.get("SQ_INSTS_VALU", 0.0)
            rate = valu * 64.0 / (s["avg_ms"] * 1e-3)
            e = {"avg_ms": s["avg_ms"], "min_ms": s["min_ms"], "launches": s["calls"], "valu_wave_instr_per_launch": valu,
                 "salu_wave_instr_per_launch": d.get("SQ_INSTS_SALU"), "lds_instr_per_launch": d.get("SQ_INSTS_LDS"),
                 "lds_bank_conflict_cycles": d.get("SQ_LDS_BANK_CONFLICT"), "wait_inst_any": d.get("SQ_WAIT_INST_ANY"),
                 "wave_cycles": d.get("SQ_WAVE_CYCLES"), "waves": d.get("SQ_WAVES"),
                 "valu_lane_ops_per_s": rate, "frac_of_valu_issue_peak": rate / PEAK}
            if "ungapped" in name and pf.get("ungapped"):
                e["cells_per_launch"] = pf["Lq"] * pf["residues"]
            if "pf_sw" in name and pf.get("gapped"):
                e["cells_per_launch"] = pf["gapped"]["cells_per_s"] * pf["gapped"]["kernel_ms"] * 1e-3
            if e.get("cells_per_launch"):
                e["valu_lane_instr_per_cell"] = valu * 64.0 / e["cells_per_launch"]
                e["cells_per_s"] = e["cells_per_launch"] / (s["avg_ms"] * 1e-3)
            summary["kernels"][short(name)] = e
            txt.append("  %-44s %8.3f ms  VALU %.3e wave-instr (SALU %.3e)  -> %.1f T lane-ops/s = %.2f of the issue peak%s"
                       % (short(name)[:44], s["avg_ms"], valu, d.get("SQ_INSTS_SALU", 0.0), rate / 1e12, rate / PEAK,
                          "; %.1f lane-instr per cell, %.2e cells/s" % (e["valu_lane_instr_per_cell"], e["cells_per_s"]) if e.get("cells_per_launch") else ""))
        txt.append("")
    txt += ["bench lines:", "  prefilter: " + json.dumps(pf), "  MAC 500:   " + json.dumps(mac), "  MAC 2000:  " + json.dumps(mac2k)]
    stats_table("mac2k")
    json.dump(summary, open(os.path.join(DST, "%s_next_rows_summary.json" % TAG), "w"), indent=1)
    open(os.path.join(DST, "%s_next_rows_summary.txt" % TAG), "w").write("\n".join(txt) + "\n")
    print("\n".join(txt))


if __name__ == "__main__":
    main()
