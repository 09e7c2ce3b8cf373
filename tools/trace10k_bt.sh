cd /tmp && export TMPDIR=/tmp
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows"
rm -rf /tmp/prof10k; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof10k -o s -- python $GRAFT_REPO_ROOT/bench.py --templates 10000 --backtrace 1 --steps 50 --warmup 5 $short > /tmp/p10k.log 2>&1
grep '^{' /tmp/p10k.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'value %.3e' % d['value'])"
f=$(find /tmp/prof10k -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-4 | cut -c1-150
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof10k/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# one step in the middle: find the stream kernels and print what lies between two of them
idx = [i for i, r in enumerate(rows) if "hhv_stream_kernel" in r["Kernel_Name"]]
a, b = idx[30], idx[31]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    print("%9.1f us  +%8.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))
PY
