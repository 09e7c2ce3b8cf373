#!/bin/bash
# The GPU-box sessions of a round, one stage per gpurun call:  gpurun -- 'bash tools/gpu_session.sh <stage> [args]'
# (what each stage produced is logged in tools/SESSIONS.md; outputs land in gpurun_out/ and the ones that are evidence are
# copied to profiles/ by hand).  Stages:
#   check [soak_s]   the driver's round-end sequence (GPU suite, default bench line, smoke) + the multi-rank bench variants
#                    + the randomised soak on HEAD (default 100 s per family)
#   ab <libdir>...   A/B of library builds on the fixed set of bench configurations (tools/gpu_ab.sh)
#   profile          rocprofv3 kernel trace + PMC passes of the headline command and its backtrace twin (tools/profile.sh)
#   next             the same for the prefilter / MAC kernels (tools/profile_next.sh)
#   mac [soak_s]     after a change of the MAC kernels: their tests, a soak of their family, 500 hits of fixed and mixed lengths, the
#                    single-wave kernels beside them (HHV_MAC_NO_PIPE=1)
#   macprof          rocprofv3 kernel statistics of the MAC kernels, dataflow and single-wave
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
stage=$1; shift
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows"
line() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('%.3e cells/s, %.2f ms/step, kernel %.2f ms (min %.2f)' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_min']))"; }
case $stage in
check)
  soak_s=${1:-100}
  timeout 1700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
  timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 6000 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
  timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
  echo "== one rank through RCCL (--force-dist), headline size"
  timeout 300 python bench.py --force-dist --steps 20 --warmup 5 $short > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err; line < $OUT/bench_force_dist.json; tail -2 $OUT/bench_force_dist.err
  echo "== configs[4] as one of 8 shards would see it: --lengths zipf --local 1, 125k templates"
  timeout 300 python bench.py --lengths zipf --local 1 --templates 125000 --steps 10 --warmup 3 $short > $OUT/bench_zipf.json 2> $OUT/bench_zipf.err; line < $OUT/bench_zipf.json
  for cfg in "--lq 431 --templates 50000" "--lq 431 --templates 50000 --backtrace 1" "--lq 512 --templates 50000" "--lq 1000 --lt 500 --templates 20000" "--lq 150 --templates 100000" "--lq 150 --templates 100000 --backtrace 1" "--lq 80 --templates 100000" "--lq 300 --templates 100000 --local 1" "--lq 300 --templates 100000 --backtrace 1" "--ss 4" "--ss 4 --backtrace 1" "--templates 10000 --backtrace 1" "--templates 20000 --backtrace 1"; do
    echo -n "== $cfg : "
    timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
  done
  echo "== soak, $soak_s s per family"
  timeout $((soak_s * 7 + 180)) python tools/soak.py $soak_s > $OUT/soak.json 2> $OUT/soak.err; cat $OUT/soak.json; tail -2 $OUT/soak.err
  echo "== the native multi-rank program, one rank through RCCL (examples/sharded_search_rccl.cpp)"
  timeout 300 ./build/sharded_search_rccl --world 1 --templates 20000 --steps 5 --check 2>&1 | tail -4
  timeout 300 ./build/sharded_search_rccl --world 1 --templates 20000 --steps 5 --backtrace 2>&1 | tail -3
  echo "== tools/scale8.sh (dry run on the GPUs visible)"
  bash tools/scale8.sh 2>&1 | tail -3
  ;;
ab)
  bash tools/gpu_ab.sh "$@"
  ;;
profile)   # profile <tag> [bench args]: raw passes, summary into gpurun_out/profiles_out, raw directories removed (size limit)
  bash tools/profile.sh "$@" > $OUT/profile_$1.log 2>&1
  HHV_PROFILE_OUT=$OUT/profiles_out python tools/summarize_profile.py $1 | tail -40
  rm -rf $OUT/prof_$1
  ;;
next)
  bash tools/profile_next.sh > $OUT/profile_next.log 2>&1
  HHV_PROFILE_OUT=$OUT/profiles_out python tools/summarize_next.py ${1:-r3} | tail -30
  rm -rf $OUT/prof_next
  ;;
rows)   # rows [libname]: the side entries of bench.py (tools/bench_rows.py: ss_modes, multi_strip, masked_round) on the resident headline set
  HHV_LIB=$ROOT/hh-suite_amd/lib/libhhviterbi_${1:-hip}.so timeout 900 python - > $OUT/rows_${1:-hip}.json 2> $OUT/rows_${1:-hip}.err <<'PY'
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
from pyhhv import capi, synth, synth_stream
import bench_rows
dev = torch.device("cuda", 0)
n, Lq, Lt, K = 100000, 300, 300, 500
qf, qtr = synth_stream.query_np(Lq, synth.PB)
rec, rec_off, Ls = synth_stream.gen_stream(torch, dev, np.arange(n), np.full(n, Lt), synth.PB)
torch.cuda.synchronize()
out = {"ss_modes": bench_rows.ss_modes(torch, capi, 0, rec, rec_off, Ls, qf, qtr, K),
       "multi_strip": bench_rows.multi_strip(torch, capi, 0, rec, rec_off, Ls, K),
       "masked_round": bench_rows.masked_round(torch, capi, 0, rec, rec_off, Ls, qf, qtr, K)}
print(json.dumps(out))
PY
  tail -3 $OUT/rows_${1:-hip}.err
  python - <<PY
import json
d = json.load(open("$OUT/rows_${1:-hip}.json"))
def walk(p, v):
    if isinstance(v, dict):
        if "cells_per_s" in v:
            print("%-70s %.3e cells/s  step %.2f ms  DP %.2f ms" % (p, v["cells_per_s"], v["ms_per_step"], v["dp_kernel_ms"]))
        for k, x in v.items():
            if k == "gpu_matches_cpu_on_sample" or k == "error" or k == "results_identical":
                print("%-70s %s" % (p + "." + k, x))
            elif isinstance(x, dict):
                walk(p + "." + k, x)
walk("", d)
print("masked_round", {k: v for k, v in d["masked_round"].items() if not isinstance(v, dict)})
PY
  ;;
mac)   # mac [soak_s]: the MAC realignment after a kernel change: its tests, a soak of its family, the 500-hit timing (fixed and mixed lengths)
  timeout 900 python -m pytest tests/test_mac.py tests/test_dropin_realign.py tests/test_pipeline.py tests/test_dropin_apps.py -q -m gpu -x 2>&1 | tail -3
  timeout 300 python tools/soak.py ${1:-40} 777 mac 2>$OUT/soak_mac.err | tail -3
  for a in "500 300 300 50" "500 300 0 50" "500 700 0 20" "2000 300 300 20"; do timeout 300 python tools/bench_mac.py $a 2>>$OUT/bench_mac.err | tee -a $OUT/bench_mac.jsonl | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('n_hits','Lq','Lt','gpu_kernels_ms','mismatches_vs_reference','checked')}, d['resident_set']['gpu_kernels_ms'])"; done
  echo "single-wave kernels (HHV_MAC_NO_PIPE=1):"
  for a in "500 300 300 0" "500 300 0 0"; do HHV_MAC_NO_PIPE=1 timeout 300 python tools/bench_mac.py $a 2>>$OUT/bench_mac.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('n_hits','Lq','Lt','gpu_kernels_ms')}, d['resident_set']['gpu_kernels_ms'])"; done
  ;;
macprof)   # kernel statistics of the MAC kernels: the wavefront pipeline and the single-wave kernels on the same 500 hits
  cd /tmp && export TMPDIR=/tmp
  for m in pipe single; do
    e=""; [ $m = single ] && e="HHV_MAC_NO_PIPE=1"
    env $e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mac_$m -o stats -- python $ROOT/tools/bench_mac.py 500 300 ${1:-300} 0 > $OUT/prof_mac_$m.txt 2>&1
    f=$(find $OUT/prof_mac_$m -name "*kernel_stats.csv" | head -1); echo "== $m"; head -9 "$f" | cut -d, -f1-7; cp "$f" $OUT/mac_${m}_kernel_stats.csv; rm -rf $OUT/prof_mac_$m
  done
  ;;
prep)   # prep [n] [tag]: rocprofv3 statistics and counters of the on-device PrepareTemplateHMM (tools/profile_prep.sh)
  bash tools/profile_prep.sh ${1:-100000} ${2:-r6} 2>&1 | tail -30
  rm -rf $OUT/prof_prep/pmc_* $OUT/prof_prep/stats
  ;;
macprofx)   # macprofx "<n Lq Lt>" ...: rocprofv3 kernel statistics of tools/bench_mac.py for each shape given (dataflow kernels only)
  cd /tmp && export TMPDIR=/tmp
  i=0
  for shape in "$@"; do
    i=$((i+1)); tag=$(echo $shape | tr ' ' '_')
    timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_macx_$i -o stats -- python $ROOT/tools/bench_mac.py $shape 0 > $OUT/prof_macx_$tag.txt 2>&1
    f=$(find $OUT/prof_macx_$i -name "*kernel_stats.csv" | head -1); echo "== $shape"; head -12 "$f" | cut -d, -f1-7; cp "$f" $OUT/macx_${tag}_kernel_stats.csv; rm -rf $OUT/prof_macx_$i
    tail -1 $OUT/prof_macx_$tag.txt | cut -c1-400
  done
  ;;
r6p)   # the round's profiles: headline + backtrace (PMC, stamped with the kernel-source hash), next rows (prefilter / MAC), prepare, the 1 M pipeline
  for spec in "r6|" "r6bt|--backtrace 1"; do
    tag=${spec%%|*}; extra=${spec#*|}
    bash tools/profile.sh $tag "$extra" > $OUT/profile_$tag.log 2>&1
    HHV_PROFILE_OUT=$OUT/profiles_out python tools/summarize_profile.py $tag | tail -24
    rm -rf $OUT/prof_$tag
  done
  bash tools/gpu_session.sh next r6
  bash tools/gpu_session.sh prep 100000 r6
  echo "== pipeline, 1 M sequences"
  timeout 900 python tools/bench_pipeline.py 1000000 20000 500 | tee $OUT/profiles_out/r6_pipeline_1M.json
  timeout 600 python tools/bench_pipeline.py 200000 10000 500 | tee $OUT/profiles_out/r6_pipeline_200k.json
  ;;
r5p)   # the round's profiles: headline, backtrace, secondary structure (hhv_ss_kernel), and the kernel statistics of a 10 k backtrace search
  for spec in "r5|" "r5bt|--backtrace 1" "r5ss|--ss 4" "r5ssbt|--ss 4 --backtrace 1"; do
    tag=${spec%%|*}; extra=${spec#*|}
    bash tools/profile.sh $tag "$extra" > $OUT/profile_$tag.log 2>&1
    k=hhv_stream_kernel; case $tag in r5ss*) k=hhv_ss_kernel;; esac
    HHV_PROFILE_KERNEL=$k HHV_PROFILE_OUT=$OUT/profiles_out python tools/summarize_profile.py $tag | tail -24
    rm -rf $OUT/prof_$tag
  done
  echo "== kernel statistics of backtrace searches over 10 000 templates"
  (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bt -o stats -- python $ROOT/bench.py --templates 10000 --backtrace 1 --steps 20 --warmup 3 $short > /tmp/prof_bt.log 2>&1)
  python - <<'PY' | tee $OUT/profiles_out/r5bt10k_kernel_stats.txt
import csv, glob
for f in glob.glob("/tmp/prof_bt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hhv" in r["Name"] or "topk" in r["Name"] or "merge" in r["Name"] or "select" in r["Name"]:
            print("%-70s calls %5s  avg %10.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  ;;
mactl)   # mactl "<n Lq Lt>": timeline of the kernels of ONE timed realignment (kernel trace: start, duration, workgroups, stream)
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof_tl; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python $ROOT/tools/bench_mac.py $1 0 > $OUT/prof_mactl.txt 2>&1
  python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_tl/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed realignment = the run of MAC kernels that holds the LAST hhv_mac_trace_kernel before the resident-set repetition
mac = [r for r in rows if "mac" in r["Kernel_Name"] or "stream_kernel" in r["Kernel_Name"]]
t0 = int(mac[0]["Start_Timestamp"])
last = None
for r in mac:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last is not None and s - last > 2000000: print("---- gap %.2f ms" % ((s - last) / 1e6))
    last = max(last or 0, e)
    name = r["Kernel_Name"].replace("hhv::", "").replace("void ", "")[:62]
    print("%9.3f ms  +%8.3f ms  wg %5d x %4d  q %s  lds %6s  %s" % ((s - t0) / 1e6, (e - s) / 1e6, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Workgroup_Size_X"]), r.get("Queue_Id", "?"), r.get("LDS_Block_Size", "?"), name))
PY
  tail -1 $OUT/prof_mactl.txt | cut -c1-300
  ;;
r6d)   # drop-in host path: the packed path records (tests), then alignment() of 20 000 resident templates with its phase timers
  timeout 900 python -m pytest tests/test_gpu_trace.py tests/test_dropin_runner.py tests/test_dropin_apps.py -q -m gpu -x 2>&1 | tail -3
  HHV_DROPIN_TIMING=1 timeout 600 python tools/bench_dropin.py ${1:-20000} 16 300 2>&1 | grep -v "^$" | tail -14
  ;;
dbg)   # dbg <args of tools/dbg_ss.py>
  timeout 120 python tools/dbg_ss.py "$@" 2>&1 | tail -30
  ;;
r5c)   # probes: LDS-DMA beyond 64 KB, integer / SDWA / packed issue rates; which secondary-structure case faults; the non-SS suites on the header-record best
  ./build/lds_dma_probe
  ./build/valu_ubench int 2>&1 | grep -E "SIMD=(2|4)" > $OUT/valu_ubench_int.txt; cat $OUT/valu_ubench_int.txt
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_queue.py tests/test_gpu_pair.py -q -m gpu -x 2>&1 | tail -4
  timeout 300 python -m pytest tests/test_gpu_ss.py -v -m gpu -x 2>&1 | grep -E "PASS|FAIL|Error|fault|passed|failed" | head -20
  ;;
r5b)   # the secondary-structure kernels as workgroups of eight wavefronts with the table in LDS (hhv_ss_kernel), best slots in the header records
  timeout 900 python -m pytest tests/test_gpu_ss.py tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_queue.py tests/test_gpu_pair.py tests/test_gpu_errors.py tests/test_dropin_runner.py -q -m gpu -x 2>&1 | tail -6
  bash tools/gpu_session.sh rows hip
  for cfg in "" "--backtrace 1" "--local 1" "--lq 431 --templates 50000" "--lengths zipf --local 1 --templates 125000"; do
    echo -n "== $cfg : "
    timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
  done
  ;;
r5a)   # round 5, first session: the device error word and the launch policy (tests), then the baseline of the new bench rows on HEAD~ kernels
  timeout 600 python -m pytest tests/test_gpu_errors.py tests/test_gpu_pair.py -q -m gpu -x 2>&1 | tail -5
  bash tools/gpu_session.sh rows hip
  ;;
*) echo "unknown stage $stage"; exit 2;;
esac
