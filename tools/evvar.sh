# the DP's timing events attached to the launches (default) against hipEventRecord on the stream (HHV_EVENT_RECORDS=1), one session:
# 10 000-template step, the kernel time both ways report, the kernels of one step; then the headline size
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows"
pr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('%.4e' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f' % d['roofline']['kernel_ms'], 'min %.4f' % d['roofline']['kernel_ms_min'])"; }
for rep in 1 2; do for m in 1 0; do
  echo -n "HHV_EVENT_RECORDS=$m 10k: "; HHV_EVENT_RECORDS=$m python bench.py --templates 10000 --steps 300 --warmup 20 $short 2>/dev/null | pr
  echo -n "HHV_EVENT_RECORDS=$m 10k backtrace: "; HHV_EVENT_RECORDS=$m python bench.py --templates 10000 --backtrace 1 --steps 200 --warmup 20 $short 2>/dev/null | pr
done; done
for m in 1 0; do echo -n "HHV_EVENT_RECORDS=$m 100k: "; HHV_EVENT_RECORDS=$m python bench.py --steps 20 --warmup 5 $short 2>/dev/null | pr; done
for m in 1 0; do echo -n "HHV_EVENT_RECORDS=$m Lq 431 50k: "; HHV_EVENT_RECORDS=$m python bench.py --lq 431 --templates 50000 --steps 20 --warmup 5 $short 2>/dev/null | pr; done
for m in 1 0; do echo -n "HHV_EVENT_RECORDS=$m Lq 1000 20k: "; HHV_EVENT_RECORDS=$m python bench.py --lq 1000 --lt 500 --templates 20000 --steps 20 --warmup 5 $short 2>/dev/null | pr; done
bash tools/trace10k.sh 2>&1 | tail -7
