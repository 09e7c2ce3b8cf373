#!/bin/bash
# tools/profile_prep.sh [n_templates] -- run ON THE GPU BOX (through gpurun) from the repo root: rocprofv3 kernel statistics and the
# HBM / SQ counters of the on-device PrepareTemplateHMM (tools/bench_prepare.py), into gpurun_out/prof_prep/.
# Counters in their own passes with --kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass).
N=${1:-100000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_prep
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/tools/bench_prepare.py $N"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_prep" --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_prep" --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_prep" --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.txt 2>&1
timeout 600 rocprofv3 --kernel-include-regex "hhv_prep" --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_lds -o pmc -- $CMD > $OUT/pmc_lds.txt 2>&1
python $ROOT/tools/summarize_prep.py ${2:-r6} $N
