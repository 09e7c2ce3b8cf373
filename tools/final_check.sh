#!/bin/bash
# what the driver runs at round end, on one box: GPU suite, default bench line, smoke; plus the side benches quoted in DESIGN.md
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 4500 $OUT/bench_default.json
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for cfg in "--lq 431 --templates 50000" "--lq 431 --templates 50000 --backtrace 1" "--lq 1000 --lt 500 --templates 20000" "--lq 150 --templates 100000" "--lq 150 --templates 100000 --backtrace 1" "--lq 80 --templates 100000" "--lq 300 --templates 100000 --local 1" "--lq 300 --templates 100000 --backtrace 1"; do
  echo -n "== $cfg : "
  timeout 200 python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3e cells/s, %.2f ms/step, kernel %.2f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done
