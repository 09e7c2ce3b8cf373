#!/bin/bash
# round 2, session 2: multi-pass variants back on plain hand-off registers - parity and A/B against the round's base
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_real_profile.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -3 > $OUT/gpu_parity2.log; cat $OUT/gpu_parity2.log
HHV_AB_LIBS="base hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 431 --templates 50000|--lq 1000 --lt 500 --templates 20000|--lq 431 --templates 50000 --backtrace 1" bash tools/gpu_ab.sh > $OUT/ab5.txt 2>&1; cat $OUT/ab5.txt
