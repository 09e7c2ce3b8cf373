#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_real_profile.py tests/test_gpu_configs.py tests/test_gpu_ss.py -q -m gpu -x 2>&1 | tail -4
for cfg in "--lq 300 --templates 100000" "--lq 300 --templates 100000 --backtrace 1" "--lq 300 --templates 100000 --local 1" "--lq 431 --templates 50000" "--lq 150 --templates 100000"; do
  echo "== $cfg"
  timeout 200 python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
timeout 400 python tools/soak.py 40 > $OUT/soak_r2.json 2> $OUT/soak_r2.err; tail -c 1500 $OUT/soak_r2.json
