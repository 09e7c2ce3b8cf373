#!/bin/bash
# round 2, session 2: hand-off pulled through the LDS crossbar and delivered late - parity and A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_real_profile.py tests/test_gpu_fullsize.py tests/test_gpu_configs.py tests/test_gpu_ss.py -q -m gpu -x 2>&1 | tail -3 > $OUT/gpu_parity6.log; cat $OUT/gpu_parity6.log
HHV_AB_LIBS="base hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000|--lq 300 --templates 100000 --backtrace 1|--lq 431 --templates 50000|--lq 150 --templates 100000" bash tools/gpu_ab.sh > $OUT/ab10.txt 2>&1; cat $OUT/ab10.txt
