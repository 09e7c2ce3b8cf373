#!/usr/bin/env python3
"""Condense gpurun_out/prof_next/ (tools/profile_next.sh) into profiles/r1_next_rows_summary.{txt,json} and copy the two
kernel-stats tables: rocprofv3 evidence for the widened rows (prefilter N3, MAC realignment N4)."""
import collections
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_next")
DST = os.path.join(ROOT, "profiles")


def last_json(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return {}


def main():
    shutil.copy(os.path.join(SRC, "prefilter", "stats_kernel_stats.csv"), os.path.join(DST, "r1_prefilter_kernel_stats.csv"))
    shutil.copy(os.path.join(SRC, "mac", "stats_kernel_stats.csv"), os.path.join(DST, "r1_mac_kernel_stats.csv"))
    rows = list(csv.DictReader(open(os.path.join(SRC, "prefilter_pmc", "pmc_counter_collection.csv"))))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[r["Kernel_Name"]] += 1
    pmc = {}
    for k, v in agg.items():
        if "hhv" not in k:
            continue
        d = {c: x / calls[k] for c, x in v.items()}
        d["launches"] = calls[k]
        d["valu_wave_instr_per_simd_cycle"] = d["SQ_INSTS_VALU"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
        pmc[k] = d
    pf, mac = last_json(os.path.join(SRC, "prefilter.txt")), last_json(os.path.join(SRC, "mac.txt"))
    json.dump({"prefilter_pmc_per_launch": pmc, "bench_prefilter": pf, "bench_mac": mac},
              open(os.path.join(DST, "r1_next_rows_summary.json"), "w"), indent=1)

    def avg(table, name):
        for r in csv.DictReader(open(os.path.join(DST, table))):
            if name in r["Name"]:
                return float(r["AverageNs"]) / 1e6
        return float("nan")
    txt = ["profiles/r1_next_rows_summary.txt -- rocprofv3 evidence for the widened rows (SURVEY.md 8f N3, N4), 1x MI355X",
           "commands: tools/profile_next.sh (rocprofv3 --kernel-trace --stats; separate --pmc pass for the prefilter kernels);"
           " summary by tools/summarize_next.py", "",
           "N3 prefilter, 1e6 sequences / 3.07e8 residues, Lq=300 (tools/bench_prefilter.py):"]
    for k, d in pmc.items():
        txt.append("  %s: VALU wave-instr/launch %.3e, LDS instr %.3e, bank conflicts %d, busy cycles/XCD %.3e, VALU instr per "
                   "SIMD-cycle %.3f" % (k.split("(")[0][10:], d["SQ_INSTS_VALU"], d["SQ_INSTS_LDS"], d["SQ_LDS_BANK_CONFLICT"],
                                        d["GRBM_GUI_ACTIVE"] / 8, d["valu_wave_instr_per_simd_cycle"]))
    txt.append("  gapless: %.3f ms kernel, %.3e cells/s; Smith-Waterman (every 10th sequence): %.3f ms, %.3e cells/s"
               % (pf["ungapped"]["kernel_ms"], pf["ungapped"]["cells_per_s"], pf["gapped"]["kernel_ms"], pf["gapped"]["cells_per_s"]))
    txt.append("  kernel stats: profiles/r1_prefilter_kernel_stats.csv (hhv_pf_ungapped_kernel avg %.2f ms, hhv_pf_sw_kernel avg %.2f ms)"
               % (avg("r1_prefilter_kernel_stats.csv", "hhv_pf_ungapped"), avg("r1_prefilter_kernel_stats.csv", "hhv_pf_sw")))
    txt += ["", "N4 MAC realignment, 500 hits Lq=300 x Lt=300 local (tools/bench_mac.py):",
            "  kernel stats: profiles/r1_mac_kernel_stats.csv (avg per launch: forward %.2f ms, backward %.2f ms, MAC DP %.2f ms, "
            "backtrace %.2f ms, mask %.2f ms)" % tuple(avg("r1_mac_kernel_stats.csv", n) for n in
                                                       ("mac_forward", "mac_backward", "mac_dp", "mac_trace", "mac_mask")),
            "  bench line: " + json.dumps(mac)]
    open(os.path.join(DST, "r1_next_rows_summary.txt"), "w").write("\n".join(txt) + "\n")
    print("\n".join(txt))


if __name__ == "__main__":
    main()
