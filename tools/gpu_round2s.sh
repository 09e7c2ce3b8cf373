#!/bin/bash
# round 2, session 2: late hand-off (64-lane variants) + DPP kept for the short-query arrays: suite, A/B, bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $OUT/gpu_suite3.log; cat $OUT/gpu_suite3.log
HHV_AB_LIBS="base hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 150 --templates 100000|--lq 80 --templates 100000|--lq 300 --templates 100000|--lq 1000 --lt 500 --templates 20000" bash tools/gpu_ab.sh > $OUT/ab12.txt 2>&1; cat $OUT/ab12.txt
