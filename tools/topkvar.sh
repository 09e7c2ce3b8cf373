# one-launch top-K / merge of small sets (HHV_TOPK_SMALL) against the multi-launch path, in one session:
#   their tests, then a 10 000-template score-only and backtrace step with the switch off / on (tools/trace10k.sh prints the kernels of one step)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_merge.py tests/test_gpu_configs.py tests/test_shard_gloo.py -q -m gpu -x 2>&1 | tail -5
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('%.3e cells/s, %.3f ms/step, kernel %.3f ms (min %.3f)' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min']))"; }
for rep in 1 2; do
for m in 0 1; do
  for cfg in "--templates 10000" "--templates 10000 --backtrace 1" "--templates 20000 --backtrace 1"; do
    echo -n "HHV_TOPK_SMALL=$m $cfg : "
    HHV_TOPK_SMALL=$m timeout 200 python bench.py $cfg --steps 200 --warmup 20 $short 2>/dev/null | line
  done
done
done
echo "== kernels of one step, switch on"
bash tools/trace10k.sh 2>&1 | tail -14
