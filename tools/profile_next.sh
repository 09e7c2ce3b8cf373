#!/bin/bash
# tools/profile_next.sh -- run ON THE GPU BOX (through gpurun) from the repo root: rocprofv3 kernel statistics and SQ counters
# of the widened rows: prefilter kernels (N3) and MAC realignment (N4), into gpurun_out/prof_next/.
# Counters in their own passes with --kernel-trace only (no sys / hip / hsa tracing next to --pmc).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_next
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PF="python $ROOT/tools/bench_prefilter.py 1000000 300 0"
MAC="python $ROOT/tools/bench_mac.py 500 300 300 0"
MAC2K="python $ROOT/tools/bench_mac.py 2000 300 300 0"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefilter -o stats -- $PF > $OUT/prefilter.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mac -o stats -- $MAC > $OUT/mac.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mac2k -o stats -- $MAC2K > $OUT/mac2k.txt 2>&1
SQ1="SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM"
timeout 600 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/prefilter_pmc -o pmc -- $PF > $OUT/prefilter_pmc.txt 2>&1
timeout 600 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/prefilter_pmc2 -o pmc -- $PF > $OUT/prefilter_pmc2.txt 2>&1
timeout 600 rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $OUT/mac_pmc -o pmc -- $MAC > $OUT/mac_pmc.txt 2>&1
timeout 600 rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/mac_pmc2 -o pmc -- $MAC > $OUT/mac_pmc2.txt 2>&1
find $OUT -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -12 "$f"; done
tail -1 $OUT/prefilter.txt; tail -1 $OUT/mac.txt; tail -1 $OUT/mac2k.txt
