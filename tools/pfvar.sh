# measurement aid: A/B of the gapless prefilter kernels (bytes / int16 pairs) in one session
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_prefilter.py tests/test_dropin_prefilter.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
for v in 0 1; do for lq in 300 200 120; do echo -n "HHV_PF_PACKED=$v Lq=$lq : "; HHV_PF_PACKED=$v timeout 300 python tools/bench_prefilter.py 1000000 $lq 2000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('gapless %.3f ms %.3e cells/s, mismatches %s' % (d['ungapped']['kernel_ms'], d['ungapped']['cells_per_s'], d.get('mismatches_vs_oracle', d.get('ungapped',{}).get('mismatches'))))"; done; done; done
