#!/bin/bash
# round 2, session 2: finalized best handed on through LDS slots - parity and A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_real_profile.py tests/test_gpu_configs.py -q -m gpu -x 2>&1 | tail -2 > $OUT/gpu_parity7.log; cat $OUT/gpu_parity7.log
HHV_AB_LIBS="base hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000|--lq 300 --templates 100000 --backtrace 1|--lq 300 --templates 200000 --lengths zipf --local 1" bash tools/gpu_ab.sh > $OUT/ab16.txt 2>&1; cat $OUT/ab16.txt
