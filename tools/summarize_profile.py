#!/usr/bin/env python3
"""Condense a gpurun_out/prof_<tag>/ directory (tools/profile.sh) into profiles/<tag>_summary.{txt,json}:
per-kernel statistics of the bench command and per-launch PMC values of the dominant kernel, with the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (HBM section) applied and stated."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
KERNEL = os.environ.get("HHV_PROFILE_KERNEL", "hhv_stream_kernel")   # (hhv_pair_kernel for the two-strip launches)


def rows(pattern):
    out = []
    for f in glob.glob(os.path.join(src, pattern), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import srchash  # noqa: E402
summary = {"tag": tag, "kernel": KERNEL, "kernel_sources_sha1": srchash.kernel_sources_sha1(srchash.VITERBI)}  # what the counters were taken on (bench.py: profile_stale)
lines = []
stats = rows("stats/**/*kernel_stats.csv")
lines.append("== rocprofv3 --kernel-trace --stats (python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs1 ...)")
lines.append("%-70s %8s %14s %14s %8s" % ("kernel", "calls", "total_ns", "avg_ns", "pct"))
for r in sorted(stats, key=lambda r: -float(r.get("TotalDurationNs", 0)))[:12]:
    name = r["Name"]
    short = name if len(name) < 70 else name[:67] + "..."
    lines.append("%-70s %8s %14s %14.0f %8.2f" % (short, r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]),
                                                 float(r["Percentage"])))
    if KERNEL in name:
        summary["stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]),
                            "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}

# per-dispatch durations from the kernel trace of the same pass: min and median next to the mean (the first launches of a
# process run slower; the driver's bench takes 20 steps after 5 warm-up steps)
dur = sorted((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in rows("stats/**/*kernel_trace.csv")
             if KERNEL in r.get("Kernel_Name", ""))
if dur:
    med = dur[len(dur) // 2] if len(dur) % 2 else 0.5 * (dur[len(dur) // 2 - 1] + dur[len(dur) // 2])
    summary["stats"].update({"median_ns": med, "min_ns_trace": dur[0], "launches_in_trace": len(dur)})
    lines.append("%s: %d launches, min %.0f ns, median %.0f ns, mean %.0f ns" % (KERNEL, len(dur), dur[0], med, sum(dur) / len(dur)))
    try:   # the bench line of the profiled run itself: its HIP-event kernel time must agree with the trace
        for l in open(os.path.join(src, "stats_bench.txt")):
            if l.startswith("{"):
                b = json.loads(l)
                summary["bench_line_of_profiled_run"] = {"ms_per_step": b["ms_per_step"], "kernel_ms_mean": b["roofline"]["kernel_ms"],
                                                         "kernel_ms_min": b["roofline"].get("kernel_ms_min"),
                                                         "kernel_ms_median": b["roofline"].get("kernel_ms_median"), "value": b["value"]}
                lines.append("bench line of the same run: %.3f ms per step, kernel (HIP events) mean %.3f / median %.3f / min %.3f ms"
                             % (b["ms_per_step"], b["roofline"]["kernel_ms"], b["roofline"].get("kernel_ms_median", 0), b["roofline"].get("kernel_ms_min", 0)))
    except Exception:
        pass

pmc = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    for r in rows(sub + "/**/*counter_collection.csv"):
        if KERNEL not in r.get("Kernel_Name", ""):
            continue
        pmc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
lines.append("")
lines.append("== PMC per launch of %s (mean over profiled launches)" % KERNEL)
pm = {k: sum(v) / len(v) for k, v in pmc.items()}
for k in sorted(pm):
    lines.append("%-28s %20.1f   (%d launches)" % (k, pm[k], len(pmc[k])))
summary["pmc_mean_per_launch"] = pm
if "FETCH_SIZE" in pm:
    # rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half
    # their size (128-B requests tallied as 64 B): doubled here, as the guide prescribes.
    fetch = pm["FETCH_SIZE"] * 1024.0 * 2.0
    summary["hbm_read_bytes_per_launch_corrected"] = fetch
    lines.append("HBM read bytes/launch  (FETCH_SIZE KiB x 1024 x 2 gfx950 correction) = %.0f" % fetch)
if "WRITE_SIZE" in pm:
    wr = pm["WRITE_SIZE"] * 1024.0
    summary["hbm_write_bytes_per_launch"] = wr
    lines.append("HBM write bytes/launch (WRITE_SIZE KiB x 1024, uncalibrated)         = %.0f" % wr)
# WRITE_SIZE calibration (tools/write_calib.hip: 15 052 800 000 bytes stored per launch, 8 or 16 bytes per lane)
CAL_BYTES = 2048 * 14700 * 64 * 8
cal = {}
for r in rows("pmc_wcal/**/*counter_collection.csv"):
    if r.get("Counter_Name") == "WRITE_SIZE":
        cal.setdefault("store16" if "store16" in r.get("Kernel_Name", "") else "store8", []).append(float(r["Counter_Value"]))
if cal:
    lines.append("")
    lines.append("== WRITE_SIZE calibration (known %d bytes per launch, pattern of the backtrace stores)" % CAL_BYTES)
    summary["write_size_calibration"] = {}
    for k, v in sorted(cal.items()):
        ratio = (sum(v) / len(v)) * 1024.0 / CAL_BYTES
        summary["write_size_calibration"][k] = {"counter_KiB": sum(v) / len(v), "true_bytes": CAL_BYTES, "counter_over_true": ratio}
        lines.append("%-8s WRITE_SIZE x 1024 / true bytes = %.3f" % (k, ratio))
    f8 = summary["write_size_calibration"].get("store8", {}).get("counter_over_true")
    if f8 and "WRITE_SIZE" in pm:
        summary["hbm_write_bytes_per_launch_calibrated"] = pm["WRITE_SIZE"] * 1024.0 / f8
        lines.append("HBM write bytes/launch calibrated with the 8-B-per-lane factor            = %.0f" % summary["hbm_write_bytes_per_launch_calibrated"])
if "SQ_INSTS_VALU" in pm:
    summary["valu_wave_instr_per_launch"] = pm["SQ_INSTS_VALU"]
if "hbm_read_bytes_per_launch_corrected" in summary:
    summary["traffic_bytes_per_launch"] = summary["hbm_read_bytes_per_launch_corrected"] + summary.get(
        "hbm_write_bytes_per_launch_calibrated", summary.get("hbm_write_bytes_per_launch", 0.0))
# on the GPU box the summary goes to gpurun_out/profiles_out (HHV_PROFILE_OUT): only gpurun_out/ travels back, and the raw
# profile directories (tens of MB per pass) would push it over the 64 MiB that are copied
dst = os.environ.get("HHV_PROFILE_OUT", os.path.join(root, "profiles"))
os.makedirs(dst, exist_ok=True)
with open(os.path.join(dst, tag + "_summary.txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
with open(os.path.join(dst, tag + "_summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
import shutil
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(dst, tag + "_rocprofv3_kernel_stats.csv"))
print("\n".join(lines))
