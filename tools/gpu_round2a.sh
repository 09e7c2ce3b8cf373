#!/bin/bash
# first GPU trip of round 2: parity of the re-written stream kernel, bench with the configs[2]/[4] lines, micro-benchmarks
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 3000 $OUT/bench_default.json; tail -5 $OUT/bench_default.err
timeout 200 python bench.py --backtrace 1 --no-cpu-baseline --steps 5 > $OUT/bench_bt.json 2> $OUT/bench_bt.err; tail -c 1500 $OUT/bench_bt.json
timeout 200 $ROOT/tools/valu_ubench > $OUT/valu_ubench_r2.txt 2>&1; tail -60 $OUT/valu_ubench_r2.txt
timeout 100 $ROOT/tools/write_calib 2>&1 | tail -4
