#!/bin/bash
# tools/scale8.sh -- the whole multi-GPU measurement in ONE command, for the day an 8-GPU node is leased:
#   bash tools/scale8.sh [max_gpus]        (default: every power of two up to the GPUs visible; one GPU = the dry run)
# For N = 1, 2, 4, 8 ranks (one process per GPU):
#   * bench.py --gpus N                       fixed lengths: 125 000 templates per GPU (N = 8 IS BASELINE configs[3], 1 M templates)
#   * bench.py --gpus N --lengths zipf        configs[4]: ONE global Zipf length vector cut by hhv_shard_plan, local mode
#     (torch.distributed backend nccl = RCCL; per rank: DP kernel, all-gather, step)
#   * build/sharded_search_rccl --world N     the native program on hhv::RcclShardedRunner (librccl directly, no torch), 20 000 templates per
#     GPU of SURVEY 8(d)'s generator, --check (the merged list = ONE GPU's top K) with NCCL_DEBUG=INFO: the transport lines RCCL prints
#     for the communicator (P2P over xGMI / SHM / NET) are kept
# Output: gpurun_out/scale8.json (one JSON document) + the raw logs next to it.  Nothing here computes a scaling efficiency:
# the per-N values are what the driver (or a reader) divides.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/scale8
mkdir -p $OUT
cd $ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
NDEV=$(python - <<'PY'
import sys
sys.path.insert(0, "hh-suite_amd")
from pyhhv import capi
print(capi.device_count())
PY
)
MAXG=${1:-$NDEV}
[ "$MAXG" -gt "$NDEV" ] && MAXG=$NDEV
short="--no-cpu-baseline --no-configs1 --no-configs2 --no-configs4 --no-next-rows --no-pipeline --no-upload --no-fast-mode --no-rows"
echo "scale8: $NDEV GPU(s) visible, running up to $MAXG" >&2
for N in 1 2 4 8; do
  [ $N -gt $MAXG ] && break
  echo "== N = $N" >&2
  timeout 900 python bench.py --gpus $N --steps 10 --warmup 3 $short > $OUT/bench_fixed_$N.json 2> $OUT/bench_fixed_$N.err
  timeout 900 python bench.py --gpus $N --lengths zipf --local 1 --steps 10 --warmup 3 $short --dump-topk $OUT/topk_zipf_ranks_$N.npy > $OUT/bench_zipf_$N.json 2> $OUT/bench_zipf_$N.err
  # the verdict of the lease, not only its times: the merged top-K of N RANKS must be the list ONE rank gets from the same N
  # shards run one after the other (--virtual-shards N: same hhv_shard_plan, same hhv_topk per shard, same hhv_merge_hits)
  PER_GPU=125000; [ $N -eq 1 ] && PER_GPU=100000   # bench.py's per-GPU default of the run above
  timeout 900 python bench.py --gpus 1 --virtual-shards $N --lengths zipf --local 1 --templates $PER_GPU --steps 2 --warmup 1 $short --dump-topk $OUT/topk_zipf_virtual_$N.npy > $OUT/bench_zipf_virtual_$N.json 2> $OUT/bench_zipf_virtual_$N.err
  NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH,P2P,NET timeout 900 ./build/sharded_search_rccl --world $N --templates $((20000 * N)) --steps 5 --check --json \
    > $OUT/native_$N.out 2> $OUT/native_$N.err
  NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,GRAPH,P2P,NET timeout 900 ./build/sharded_search_rccl --world $N --templates $((20000 * N)) --steps 5 --backtrace --json \
    > $OUT/native_bt_$N.out 2> $OUT/native_bt_$N.err
done
python - "$OUT" "$MAXG" <<'PY' > $ROOT/gpurun_out/scale8.json
import json, os, re, sys
out, maxg = sys.argv[1], int(sys.argv[2])
def last_json(path):
    try:
        for line in reversed(open(path).read().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    except Exception as e:
        return {"error": repr(e)}
    return {"error": "no JSON line in %s" % os.path.basename(path)}
def transport(path):
    """the lines RCCL's log has about how the ranks of the communicator reach each other"""
    try:
        txt = open(path).read() + open(path.replace(".err", ".out")).read()
    except Exception:
        return []
    keep = [l.strip() for l in txt.splitlines() if re.search(r"via (P2P|SHM|NET|direct)|Channel \d+.*: \d+\[|Connected all|comm 0x.*nranks|NCCL version|RCCL version|xGMI|XGMI", l)]
    return keep[:40]
doc = {"what": "tools/scale8.sh: bench.py --gpus N (RCCL through torch.distributed) and build/sharded_search_rccl --world N (librccl directly) for N = 1, 2, 4, 8",
       "gpus_run": [], "runs": {}}
for n in (1, 2, 4, 8):
    if n > maxg:
        break
    doc["gpus_run"].append(n)
    e = {}
    for key, f in (("bench_fixed", "bench_fixed_%d.json"), ("bench_zipf", "bench_zipf_%d.json")):
        b = last_json(os.path.join(out, f % n))
        e[key] = {k: b.get(k) for k in ("value", "unit", "n_gpus", "ms_per_step", "templates_per_s", "per_rank", "error")} if "error" not in b else b
        if "config" in b:
            e[key]["workload"] = b["config"].get("workload")
            e[key]["shards"] = b["config"].get("shards")
    try:
        import numpy as np
        a, b = np.load(os.path.join(out, "topk_zipf_ranks_%d.npy" % n)), np.load(os.path.join(out, "topk_zipf_virtual_%d.npy" % n))
        e["merged_topk_equals_virtual_shards_on_one_gpu"] = bool(a.shape == b.shape and a.dtype == b.dtype and np.ascontiguousarray(a).tobytes() == np.ascontiguousarray(b).tobytes())
        e["merged_topk_records"] = int(a.shape[0])
    except Exception as ex:
        e["merged_topk_equals_virtual_shards_on_one_gpu"] = "not compared: %r" % (ex,)
    e["native_rccl"] = last_json(os.path.join(out, "native_%d.out" % n))
    e["native_rccl_backtrace"] = last_json(os.path.join(out, "native_bt_%d.out" % n))
    e["rccl_transport_lines"] = transport(os.path.join(out, "native_%d.err" % n))
    doc["runs"]["N%d" % n] = e
print(json.dumps(doc, indent=1))
PY
python - <<'PY'
import json
d = json.load(open("gpurun_out/scale8.json"))
for n, e in d["runs"].items():
    bf, bz, nr = e["bench_fixed"], e["bench_zipf"], e["native_rccl"]
    print(n, "bench fixed %s cells/s | zipf %s (top-K = virtual shards: %s) | native %s (check %s, identical %s) | transport lines %d" % (
        bf.get("value"), bz.get("value"), e.get("merged_topk_equals_virtual_shards_on_one_gpu"), nr.get("cells_per_s"), nr.get("check_one_gpu"),
        nr.get("merged_identical_on_all_ranks"), len(e["rccl_transport_lines"])))
PY
