#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/profile.sh r2 "" > /dev/null 2>&1
bash tools/profile.sh r2bt "--backtrace 1" > /dev/null 2>&1
python tools/summarize_profile.py r2 | tail -32
python tools/summarize_profile.py r2bt | tail -32
mkdir -p gpurun_out/profiles_out && cp profiles/r2_summary.* profiles/r2bt_summary.* gpurun_out/profiles_out/
cp gpurun_out/prof_r2/stats/*kernel_stats.csv gpurun_out/profiles_out/r2_rocprofv3_kernel_stats.csv 2>/dev/null
cp gpurun_out/prof_r2bt/stats/*kernel_stats.csv gpurun_out/profiles_out/r2bt_rocprofv3_kernel_stats.csv 2>/dev/null
ls gpurun_out/prof_r2/stats | head
