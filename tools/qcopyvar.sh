# the query's way to the device (HHV_QUERY_COPY=1: copy operation + event in hhv_set_query; default: one upload kernel launched by
# hhv_align_async that also sets the ticket counter), A/B in one session on a 10 000-template step + the kernels of one step
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows"
for rep in 1 2; do for m in 1 0; do echo -n "HHV_QUERY_COPY=$m: "; HHV_QUERY_COPY=$m python bench.py --templates 10000 --steps 300 --warmup 20 $short 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('%.4e' % d['value'], '%.4f' % d['ms_per_step'], d['roofline']['kernel_ms'])"; done; done
bash tools/trace10k.sh 2>&1 | tail -8
