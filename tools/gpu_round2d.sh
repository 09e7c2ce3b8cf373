#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
timeout 300 python tools/bench_mac.py 500 300 300 8 2>&1 | tail -1
timeout 600 python tools/bench_apps.py 10000 32 8 2>&1 | tail -1
