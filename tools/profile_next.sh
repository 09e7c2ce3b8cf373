#!/bin/bash
# tools/profile_next.sh -- run ON THE GPU BOX (through gpurun) from the repo root: rocprofv3 kernel statistics of the
# widened rows: prefilter kernels (N3) and MAC realignment (N4), into gpurun_out/prof_next/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_next
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefilter -o stats -- python $ROOT/tools/bench_prefilter.py 1000000 300 0 > $OUT/prefilter.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mac -o stats -- python $ROOT/tools/bench_mac.py 500 300 300 0 > $OUT/mac.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/prefilter_pmc -o pmc -- python $ROOT/tools/bench_prefilter.py 1000000 300 0 > $OUT/prefilter_pmc.txt 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -12 "$f"; done
tail -1 $OUT/prefilter.txt; tail -1 $OUT/mac.txt
