#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 900 python -m pytest tests/test_mac.py tests/test_dropin_realign.py tests/test_pipeline.py -q -m gpu -x 2>&1 | tail -8
timeout 300 python tools/bench_mac.py 500 300 300 8 2>&1 | tail -1
timeout 300 python tools/bench_mac.py 500 300 0 8 2>&1 | tail -1
timeout 300 python tools/bench_mac.py 2000 300 300 0 2>&1 | tail -1 | cut -c1-400
