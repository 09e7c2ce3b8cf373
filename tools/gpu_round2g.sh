#!/bin/bash
# round 2, session 2: parity of the in-place hand-off / unrolled step loop, A/B of its variants, dependent-chain ubench
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
./build/mix_ubench > $OUT/mix_ubench2.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_configs.py tests/test_real_profile.py tests/test_gpu_ss.py -q -m gpu -x 2>&1 | tail -5 > $OUT/gpu_parity.log; cat $OUT/gpu_parity.log
HHV_AB_LIBS="base hip v2 v3 v4" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000" bash tools/gpu_ab.sh > $OUT/ab1.txt 2>&1
HHV_AB_LIBS="base hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000 --backtrace 1|--lq 431 --templates 50000|--lq 300 --templates 100000 --local 1" bash tools/gpu_ab.sh > $OUT/ab2.txt 2>&1
cat $OUT/ab1.txt $OUT/ab2.txt
