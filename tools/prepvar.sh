# measurement aid: tools/bench_prepare.py under launch variants of the fused prepare kernel
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_prepare.py tests/test_pipeline.py -q -m gpu -x 2>&1 | tail -2
run() { echo -n "$* : "; env "$@" timeout 300 python tools/bench_prepare.py 100000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(round(d['prepare_kernel_ms'],3), d['records_match_oracle'])"; }
run A=1
run A=1
for w in 4 5 6; do for a in 0 1; do run HHV_PREP_WAVES=$w HHV_PREP_W3ALL=$a; done; done
