#!/bin/bash
# round 2, session 2: where do the waves stall?  early profile reads (v5), no wait for the profile (v6, wrong results,
# timing only), one wave per SIMD (HHV_BLOCKS_PER_CU=4)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
HHV_AB_LIBS="hip v5 v6" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000" bash tools/gpu_ab.sh > $OUT/ab3.txt 2>&1
echo "one wave per SIMD:" >> $OUT/ab3.txt
HHV_BLOCKS_PER_CU=4 HHV_AB_LIBS="hip" HHV_AB_REPS=1 HHV_AB_CFGS="--lq 300 --templates 100000" bash tools/gpu_ab.sh >> $OUT/ab3.txt 2>&1
echo "six waves per CU:" >> $OUT/ab3.txt
HHV_BLOCKS_PER_CU=6 HHV_AB_LIBS="hip" HHV_AB_REPS=1 HHV_AB_CFGS="--lq 300 --templates 100000" bash tools/gpu_ab.sh >> $OUT/ab3.txt 2>&1
cat $OUT/ab3.txt
