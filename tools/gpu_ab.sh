#!/bin/bash
# A/B of several builds of the library in one session (alternating, so that clock drift hits all): HHV_LIB selects the .so
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
LIBS=${HHV_AB_LIBS:-"base hip"}
CFGS=${HHV_AB_CFGS:-"--lq 300 --templates 100000"}
for rep in $(seq 1 ${HHV_AB_REPS:-3}); do
  IFS='|' read -ra CF <<< "$CFGS"
  for cfg in "${CF[@]}"; do
    for l in $LIBS; do
      lib=$ROOT/hh-suite_amd/lib/libhhviterbi_$l.so
      echo -n "$l $cfg : "
      HHV_LIB=$lib timeout 200 python bench.py $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print(round(d['ms_per_step'],3), round(d['roofline']['kernel_ms'],3), round(d['roofline']['kernel_ms_min'],3))"
    done
  done
done
