#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -25 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
for cfg in "--lq 150 --templates 100000" "--lq 150 --templates 100000 --backtrace 1" "--lq 80 --templates 100000" "--lq 64 --templates 100000" "--lq 300 --templates 100000" "--lq 300 --templates 100000 --backtrace 1" "--lq 431 --templates 50000 --backtrace 1"; do
  echo "== $cfg"
  timeout 200 python bench.py $cfg --steps 5 --warmup 2 --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
