#!/bin/bash
# round 2, session 2: late hand-off, all variants - full GPU suite and A/B incl. local mode and the short-query arrays
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $OUT/gpu_suite2.log; cat $OUT/gpu_suite2.log
HHV_AB_LIBS="base hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000 --local 1|--lq 300 --templates 100000 --local 1 --backtrace 1|--lq 150 --templates 100000|--lq 80 --templates 100000|--lq 300 --templates 100000" bash tools/gpu_ab.sh > $OUT/ab11.txt 2>&1; cat $OUT/ab11.txt
