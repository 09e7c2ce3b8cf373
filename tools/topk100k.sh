timeout 900 python -m pytest tests/test_gpu_merge.py tests/test_gpu_configs.py tests/test_gpu_topk_fuzz.py -q -m gpu -x 2>&1 | tail -3
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows"
pr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('%.4e' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f' % d['roofline']['kernel_ms'], 'min %.4f' % d['roofline']['kernel_ms_min'], 'step-kernel %.1f us' % (1e3*(d['ms_per_step']-d['roofline']['kernel_ms'])))"; }
for rep in 1 2; do for m in 0 1; do
  echo -n "HHV_TOPK_SMALL=$m 100k: "; HHV_TOPK_SMALL=$m python bench.py --steps 30 --warmup 5 $short 2>/dev/null | pr
  echo -n "HHV_TOPK_SMALL=$m 100k backtrace: "; HHV_TOPK_SMALL=$m python bench.py --backtrace 1 --steps 20 --warmup 5 $short 2>/dev/null | pr
done; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof100k; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof100k -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $short > /tmp/p100k.log 2>&1
f=$(find /tmp/prof100k -name "*kernel_stats.csv" | head -1); grep hhv $f | cut -d, -f1-4 | cut -c1-130
