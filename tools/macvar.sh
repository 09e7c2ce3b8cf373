# measurement aid: tools/bench_mac.py shapes under environment variants (MAC launch policy)
cd $GRAFT_REPO_ROOT
run() { echo -n "== $* : "; env "$@" timeout 300 python tools/bench_mac.py $SHAPE 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['gpu_kernels_ms'], d['resident_set']['gpu_kernels_ms'])"; }
for SHAPE in "500 300 300" "500 300 0" "500 700 0"; do
  echo "#### $SHAPE"
  run A=1
  run GPU_MAX_HW_QUEUES=16
done
for SHAPE in "2000 300 300" "2000 300 0" "1000 300 300"; do
  echo "#### $SHAPE"
  run A=1
  run HHV_MAC_DF_HITS=0
  run HHV_MAC_DF_HITS=256
  run HHV_MAC_DF_HITS=384
done
