#!/bin/bash
# round 2, session 2: how fast is a LONE wave of the real kernel?  1, 2, 4, 8 single-wave workgroups per CU
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for b in 1 2 3 4 5 6 8; do
  echo -n "HHV_BLOCKS_PER_CU=$b : "
  HHV_BLOCKS_PER_CU=$b timeout 200 python bench.py --lq 300 --templates 100000 --steps 4 --warmup 1 --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3))"
done > $OUT/blocks_per_cu.txt 2>&1
cat $OUT/blocks_per_cu.txt
