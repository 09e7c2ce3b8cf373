#!/bin/bash
# round 2, session 2: shader-clock totals of the sections of a step (measurement build v7 = -DHHV_EXP_TIMING)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for cfg in "--lq 300 --templates 100000" "--lq 300 --templates 100000 --local 1"; do
HHV_DEBUG_CLK=1 HHV_LIB=$ROOT/hh-suite_amd/lib/libhhviterbi_v7.so timeout 200 python bench.py $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d.get('debug_clk'))"
done > $OUT/timing.txt 2>&1
cat $OUT/timing.txt
