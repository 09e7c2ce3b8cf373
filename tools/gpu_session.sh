#!/bin/bash
# The GPU-box sessions of a round, one stage per gpurun call:  gpurun -- 'bash tools/gpu_session.sh <stage> [args]'
# (what each stage produced is logged in tools/SESSIONS.md; outputs land in gpurun_out/ and the ones that are evidence are
# copied to profiles/ by hand).  Stages:
#   check [soak_s]   the driver's round-end sequence (GPU suite, default bench line, smoke) + the multi-rank bench variants
#                    + the randomised soak on HEAD (default 100 s per family)
#   ab <libdir>...   A/B of library builds on the fixed set of bench configurations (tools/gpu_ab.sh)
#   profile          rocprofv3 kernel trace + PMC passes of the headline command and its backtrace twin (tools/profile.sh)
#   next             the same for the prefilter / MAC kernels (tools/profile_next.sh)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
stage=$1; shift
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-upload --no-fast-mode"
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
  for cfg in "--lq 431 --templates 50000" "--lq 431 --templates 50000 --backtrace 1" "--lq 512 --templates 50000" "--lq 1000 --lt 500 --templates 20000" "--lq 150 --templates 100000" "--lq 150 --templates 100000 --backtrace 1" "--lq 80 --templates 100000" "--lq 300 --templates 100000 --local 1" "--lq 300 --templates 100000 --backtrace 1"; do
    echo -n "== $cfg : "
    timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
  done
  echo "== soak, $soak_s s per family"
  timeout $((soak_s * 7 + 180)) python tools/soak.py $soak_s > $OUT/soak.json 2> $OUT/soak.err; cat $OUT/soak.json; tail -2 $OUT/soak.err
  echo "== the native multi-rank program, one rank through RCCL (examples/sharded_search_rccl.cpp)"
  timeout 300 ./build/sharded_search_rccl --world 1 --templates 20000 --steps 5 --check 2>&1 | tail -4
  timeout 300 ./build/sharded_search_rccl --world 1 --templates 20000 --steps 5 --backtrace 2>&1 | tail -3
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
r4s)   # chains of pair launches for queries of more than two strips: parity, then HHV_PAIR=0 (one launch per strip) against the default
  timeout 500 python -m pytest tests/test_gpu_pair.py tests/test_gpu_lengths.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -6
  for cfg in "--lq 700 --templates 30000" "--lq 1000 --lt 500 --templates 20000" "--lq 1280 --lt 500 --templates 16000" "--lq 2000 --lt 500 --templates 10000" "--lq 1000 --lt 500 --templates 20000 --backtrace 1" "--lq 1000 --lt 500 --templates 20000 --local 1"; do
    for pv in 0 x 0 x; do
      echo -n "HHV_PAIR=$pv $cfg : "
      if [ $pv = x ]; then timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line; else HHV_PAIR=$pv timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line; fi
    done
  done
  ;;
r4r)   # hipcc scheduling strategy max-ilp (lib "ilp") against the default build
  for cfg in "" "--backtrace 1" "--local 1" "--lq 150 --templates 100000" "--lq 512 --templates 50000" "--lengths zipf --local 1 --templates 125000"; do
    for lib in hip ${VARIANT:-ilp} hip ${VARIANT:-ilp}; do
      echo -n "$lib $cfg : "
      HHV_LIB=$ROOT/hh-suite_amd/lib/libhhviterbi_$lib.so timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
    done
  done
  ;;
r4q)   # the multi-strip configurations on the default build
  for cfg in "--lq 512 --templates 50000" "--lq 640 --templates 50000" "--lq 431 --templates 50000" "--lq 431 --templates 50000 --backtrace 1" "--lq 512 --templates 50000 --backtrace 1" "--lq 1000 --lt 500 --templates 20000" "--lq 2000 --lt 500 --templates 10000"; do
    echo -n "$cfg : "
    timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
  done
  ;;
r4p)   # multi-strip step loops unrolled by two like the single-strip ones (-DHHV_EXP_MULTI_UNROLL, lib "mu"): A/B, then parity on mu
  for cfg in "--lq 512 --templates 50000" "--lq 640 --templates 50000" "--lq 431 --templates 50000" "--lq 431 --templates 50000 --backtrace 1" "--lq 1000 --lt 500 --templates 20000" "--lq 512 --templates 50000 --backtrace 1"; do
    for lib in hip mu hip mu; do
      echo -n "$lib $cfg : "
      HHV_LIB=$ROOT/hh-suite_amd/lib/libhhviterbi_$lib.so timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
    done
  done
  HHV_LIB=$ROOT/hh-suite_amd/lib/libhhviterbi_mu.so timeout 600 python -m pytest tests/test_gpu_pair.py tests/test_gpu_parity.py tests/test_gpu_lengths.py -q -m gpu -x 2>&1 | tail -4
  ;;
r4j)   # timing-only builds (WRONG results): pair kernels without waits (pns), without waits and FIFO traffic (pnf), multi-pass bodies without carry traffic (mnc)
  for cfg in "--lq 512 --templates 50000" "--lq 431 --templates 50000"; do
    for lib in hip pns pnf mnc; do for pv in 1 0; do
      [ $pv = 0 ] && [ $lib != mnc ] && [ $lib != hip ] && continue
      echo -n "$lib HHV_PAIR=$pv $cfg : "
      HHV_LIB=$ROOT/hh-suite_amd/lib/libhhviterbi_$lib.so HHV_PAIR=$pv timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
    done; done
  done
  ;;
r4i)   # where the multi-pass penalty sits: per-launch durations of the two passes (kernel trace), single pass at the same set size
  for cfg in "--lq 256 --templates 50000" "--lq 320 --templates 50000"; do echo -n "$cfg : "; timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line; done
  for pv in 0 1; do for lq in 512 640; do
    echo "== HHV_PAIR=$pv --lq $lq --templates 50000: every dispatch of the stream / pair kernels (us)"
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_kt && HHV_PAIR=$pv timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_kt -o kt -- python $ROOT/bench.py --lq $lq --templates 50000 --steps 4 --warmup 1 $short > /tmp/prof_kt.log 2>&1)
    python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_kt/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "hhv_stream_kernel" in r["Kernel_Name"] or "hhv_pair_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print("  ".join("%s:%.0f" % (r["Kernel_Name"].split("<")[1].split(">")[0].replace(" ", "")[:18], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows[-8:]))
PY
  done; done
  ;;
r4h)   # what a step costs at R rows per lane, single pass against multi pass (is the multi-pass penalty VALU or latency?)
  for cfg in "--lq 320 --templates 100000" "--lq 256 --templates 100000" "--lq 192 --templates 100000" "--lq 640 --templates 50000" "--lq 512 --templates 50000" "--lq 384 --templates 50000"; do for pv in 0 1; do
    echo -n "HHV_PAIR=$pv $cfg : "
    HHV_PAIR=$pv timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
  done; done
  ;;
r4g)   # two-strip queries as ONE launch of two-wave workgroups (hhv_pair_kernel): parity, then HHV_PAIR=0 / 1 on the bench configurations
  timeout 900 python -m pytest tests/test_gpu_pair.py -q -m gpu -x 2>&1 | tail -12
  for rep in 1 2; do for cfg in "--lq 431 --templates 50000" "--lq 431 --templates 50000 --backtrace 1" "--lq 512 --templates 50000" "--lq 640 --templates 40000" "--lq 350 --templates 50000 --backtrace 1" "--lq 431 --templates 100000 --lengths zipf --local 1"; do for pv in 0 1; do
    echo -n "HHV_PAIR=$pv $cfg : "
    HHV_PAIR=$pv timeout 200 python bench.py $cfg --steps 10 --warmup 3 $short 2>/dev/null | line
  done; done; done
  ;;
r4f)   # carry rows of the multi-pass variants in blocks (lanes 0..31 load a chunk of rows ahead, lane 0 takes its row by v_readlane): parity + A/B
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_lengths.py tests/test_gpu_queue.py -q -m gpu 2>&1 | tail -5
  HHV_AB_LIBS="nocb hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 431 --templates 50000|--lq 431 --templates 50000 --backtrace 1|--lq 700 --templates 30000|--lq 1000 --lt 500 --templates 20000|--lq 2000 --lt 300 --templates 10000 --local 1" bash tools/gpu_ab.sh
  ;;
r4e)   # trace kernel specialised on (R, encoding): parity + kernel trace; then the round's profiles (r4, r4bt)
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_configs.py tests/test_gpu_queue.py tests/test_gpu_ss.py tests/test_gpu_lengths.py -q -m gpu 2>&1 | tail -5
  for n in 100000 10000; do
    echo "== backtrace searches over $n templates"
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bt -o stats -- python $ROOT/bench.py --lq 300 --templates $n --backtrace 1 --steps 5 --warmup 2 $short > /tmp/prof_bt.log 2>&1)
    python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_bt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hhv" in r["Name"] or "topk" in r["Name"] or "merge" in r["Name"]:
            print("%-60s calls %5s  avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
  for tag in r4 r4bt; do
    extra=""; [ $tag = r4bt ] && extra="--backtrace 1"
    bash tools/profile.sh $tag "$extra" > $OUT/profile_$tag.log 2>&1
    HHV_PROFILE_OUT=$OUT/profiles_out python tools/summarize_profile.py $tag | tail -34
    rm -rf $OUT/prof_$tag
  done
  ;;
r4d)   # trace speculation on / off at 10 k and 100 k, scorr in registers, top-K select with early exit: parity + kernel trace
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adversarial.py tests/test_gpu_merge.py tests/test_gpu_configs.py tests/test_gpu_queue.py tests/test_gpu_ss.py tests/test_gpu_runner.py tests/test_gpu_fullsize.py -q -m gpu 2>&1 | tail -8
  for spec in 0 1; do for n in 100000 10000; do
    echo "== HHV_TRACE_SPECULATE=$spec, backtrace searches over $n templates"
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bt && HHV_TRACE_SPECULATE=$spec timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bt -o stats -- python $ROOT/bench.py --lq 300 --templates $n --backtrace 1 --steps 5 --warmup 2 $short > /tmp/prof_bt.log 2>&1)
    python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_bt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hhv" in r["Name"] or "topk" in r["Name"] or "merge" in r["Name"]:
            print("%-60s calls %5s  avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done; done
  ;;
r4c)   # scorr kernel with pipelined tile loads, pcm 3 on the device, the default bench line with the 8(d) generator
  timeout 600 python -m pytest tests/test_prepare.py tests/test_gpu_errors.py tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -8
  timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 7000 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
  for n in 100000 10000; do
    echo "== kernel trace of backtrace searches over $n templates"
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bt -o stats -- python $ROOT/bench.py --lq 300 --templates $n --backtrace 1 --steps 5 --warmup 2 $short > /tmp/prof_bt.log 2>&1; grep -c . /tmp/prof_bt.log > /dev/null)
    python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_bt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hhv" in r["Name"] or "topk" in r["Name"] or "merge" in r["Name"]:
            print("%-60s calls %5s  avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
  ;;
r4b)   # the tests behind the one that stopped r4a + the kernel trace of the backtrace step
  timeout 1200 python -m pytest tests/test_gpu_lengths.py tests/test_gpu_merge.py tests/test_gpu_parity.py tests/test_gpu_queue.py tests/test_gpu_rccl_example.py tests/test_gpu_runner.py tests/test_gpu_ss.py tests/test_layout.py tests/test_mac.py tests/test_pipeline.py tests/test_prefilter.py tests/test_prepare.py tests/test_real_profile.py tests/test_gpu_errors.py -q -m gpu 2>&1 | tail -25
  for n in 100000 10000; do
    echo "== kernel trace of backtrace searches over $n templates"
    (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bt -o stats -- python $ROOT/bench.py --lq 300 --templates $n --backtrace 1 --steps 5 --warmup 2 $short > /tmp/prof_bt.log 2>&1; tail -2 /tmp/prof_bt.log)
    python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_bt/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hhv" in r["Name"] or "topk" in r["Name"] or "merge" in r["Name"]:
            print("%-60s calls %5s  avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  done
  ;;
r4a)   # round 4, first kernel session: the whole GPU suite on the new tests / top-K / trace chain, the NaN probe, A/B base - nosign - hip
  timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
  ./build/nan_probe
  HHV_AB_LIBS="base nosign hip" HHV_AB_REPS=2 HHV_AB_CFGS="--lq 300 --templates 100000 --backtrace 1|--lq 300 --templates 10000 --backtrace 1|--lq 300 --templates 100000" bash tools/gpu_ab.sh
  echo "== kernel trace of one backtrace search over 100 k templates"
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bt -o stats -- python $ROOT/bench.py --lq 300 --templates 100000 --backtrace 1 --steps 5 --warmup 2 $short > /dev/null 2>&1
  python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/prof_bt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        if "hhv" in r["Name"] or "topk" in r["Name"] or "merge" in r["Name"]:
            print("%-60s calls %5s  avg %10.1f us  total %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
PY
  ;;
*) echo "unknown stage $stage"; exit 2;;
esac
