#!/bin/bash
# round 2, session 2: what clock does the chip run the stream kernel at?  (rocm-smi samples during a long run; the same for the VALU ubench)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
(timeout 120 python bench.py --steps 1500 --warmup 3 --no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows > $OUT/long_bench.json 2>/dev/null) &
BP=$!
sleep 8
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i -E "sclk|mclk|power|junction|fclk" | tr '\n' ';'; echo; sleep 1.5; done > $OUT/smi_kernel.txt
wait $BP
python -c "import json; d=json.load(open('$OUT/long_bench.json')); print('long run', d['ms_per_step'], d['roofline']['kernel_ms'])" >> $OUT/smi_kernel.txt
(for i in 1 2 3 4 5 6; do ./build/mix_ubench > /dev/null 2>&1; done) &
UP=$!
sleep 3
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|power" | tr '\n' ';'; echo; sleep 1; done > $OUT/smi_ubench.txt
kill $UP 2>/dev/null
cat $OUT/smi_kernel.txt $OUT/smi_ubench.txt
