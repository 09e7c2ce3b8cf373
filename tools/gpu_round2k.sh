#!/bin/bash
# round 2, session 2: parity after the in-place hand-off + windowed backtrace, backtrace timing, MFMA product probe
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
./build/mfma_probe > $OUT/mfma_probe.txt 2>&1; cat $OUT/mfma_probe.txt
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
HHV_AB_LIBS="base hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000 --backtrace 1|--lq 431 --templates 50000|--lq 300 --templates 100000" bash tools/gpu_ab.sh > $OUT/ab4.txt 2>&1; cat $OUT/ab4.txt
