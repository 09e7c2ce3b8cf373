ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT/prof_next
timeout 400 python -m pytest tests -q -m gpu -x 2>&1 | tail -3 > $OUT/gpu_suite.log; cat $OUT/gpu_suite.log
timeout 200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
(cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_next/mac -o stats -- python $ROOT/tools/bench_mac.py 500 300 300 0 > $OUT/prof_next/mac.txt 2>&1)
tail -1 $OUT/prof_next/mac.txt | cut -c1-400
head -8 $OUT/prof_next/mac/stats_kernel_stats.csv
