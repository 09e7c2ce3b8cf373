#!/bin/bash
# tools/profile.sh -- run ON THE GPU BOX (through gpurun) from the repo root: collects the rocprofv3
# kernel statistics and the PMC counters of the bench workload into gpurun_out/prof_<tag>/.
# Counters are collected only for the engine's kernels (--kernel-include-regex hhv_: bench.py's synthetic database is made by
# ~2 * 10^5 tiny torch launches, and counter collection on every one of them crashed rocprofv3), in separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots"); no sys/hip/hsa tracing is combined with --pmc.
TAG=${1:-r3}
EXTRA=${2:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows $EXTRA"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH --steps 20 --warmup 5 > $OUT/stats_bench.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_" --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH --steps 2 --warmup 0 > $OUT/pmc_fetch.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_" --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH --steps 2 --warmup 0 > $OUT/pmc_write.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_" --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- $BENCH --steps 2 --warmup 0 > $OUT/pmc_sq.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_" --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc_lds -o pmc -- $BENCH --steps 2 --warmup 0 > $OUT/pmc_lds.txt 2>&1
# WRITE_SIZE calibration on a known byte count in the backtrace store pattern (tools/write_calib.hip)
if [ -x $ROOT/tools/write_calib ]; then
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_wcal -o pmc -- $ROOT/tools/write_calib > $OUT/pmc_wcal.txt 2>&1
fi
find $OUT -name "*.csv" | head -30
