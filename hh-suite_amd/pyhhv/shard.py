"""Template-database sharding across the GPUs of one node and the top-K exchange.

One process per GPU.  Every template is independent of every other (the reference treats SIMD
batches as independent OpenMP iterations, /root/reference src/hhviterbirunner.cpp:122), so the DP
itself needs no communication: each rank aligns its own shard.  The only exchange is the final hit
list: every rank selects its K best records on the device (hhv_topk), ONE all_gather moves K fixed
size records per rank (K=500 -> 20 KB per rank; latency bound on xGMI), and every rank performs the
same deterministic merge (score descending, ties by global template id ascending).

This module holds only the partitioning and the exchange/merge logic; it works on any
torch.distributed backend ("nccl" = RCCL on the GPUs, "gloo" in the CPU tests).
"""
import numpy as np

REC_I32 = 10  # one hhv_hit record = 10 x 4 bytes: score, viterbi_score, score_ss, index, i1, j1, i2, j2, nsteps, matched_cols
COL_INDEX = 3


def shard_templates(lengths, world):
    """Partition template ids over `world` ranks.

    Templates are sorted by length descending (as the reference does before batching,
    src/hhviterbirunner.cpp:117-119), cut into bins of 64 consecutive templates and the bins are
    assigned greedily to the currently least loaded rank (LPT on sum of L+1 = stream records, the
    unit of DP work).  Equal lengths degenerate to contiguous N/world blocks.
    Returns a list of int64 arrays of global template ids, each sorted by length descending."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = lengths.shape[0]
    order = np.argsort(-lengths, kind="stable")
    if n == 0:
        return [np.zeros(0, dtype=np.int64) for _ in range(world)]
    if lengths.min() == lengths.max():
        cuts = [(n * r) // world for r in range(world + 1)]
        return [order[cuts[r]:cuts[r + 1]] for r in range(world)]
    bsz = max(1, min(64, n // (4 * world)))   # 64-template bins, smaller for tiny databases
    bins = [order[a:a + bsz] for a in range(0, n, bsz)]
    load = np.zeros(world, dtype=np.int64)
    parts = [[] for _ in range(world)]
    for b in bins:  # bins are already in descending work order
        r = int(np.argmin(load))
        parts[r].append(b)
        load[r] += int((lengths[b] + 1).sum())
    return [np.concatenate(p) if p else np.zeros(0, dtype=np.int64) for p in parts]


def merge_records(torch, records, K):
    """records: (m, 10) int32 tensor of hhv_hit records whose `index` field already holds GLOBAL template
    ids (invalid padding records carry index < 0).  Returns the K best, sorted by score descending,
    ties by global id ascending -- identical on every rank."""
    score = records[:, 0].contiguous().view(torch.float32)
    gid = records[:, COL_INDEX].to(torch.int64)
    valid = gid >= 0
    # composite ordering: primary score desc, secondary gid asc (stable sorts, secondary key first)
    big = torch.iinfo(torch.int64).max
    o1 = torch.argsort(torch.where(valid, gid, torch.full_like(gid, big)), stable=True)
    s1 = torch.where(valid, score, torch.full_like(score, float("-inf")))[o1]
    o2 = torch.argsort(s1, descending=True, stable=True)
    order = o1[o2]
    nvalid = int(valid.sum().item())
    return records[order[:min(K, nvalid)]]


def exchange_and_merge(torch, dist, local_records, K, group=None):
    """local_records: (K, 10) int32 tensor (device of the backend), global ids in the index column, padding = -1.
    ONE all_gather of K records per rank, then the common merge."""
    world = dist.get_world_size(group) if dist is not None and dist.is_initialized() else 1
    if world == 1:
        return merge_records(torch, local_records, K)
    src = local_records.contiguous()
    if src.is_cuda and dist.get_backend(group) == "gloo":   # debug path only (see bench.py HHV_BENCH_BACKEND)
        src = src.cpu()
    gathered = torch.empty((world * K, REC_I32), dtype=torch.int32, device=src.device)
    dist.all_gather_into_tensor(gathered, src, group=group)
    return merge_records(torch, gathered.to(local_records.device), K)


def to_global_ids(torch, records, global_ids):
    """Replace the local template index by the global id; padding (0xFF.. records) -> -1."""
    idx = records[:, COL_INDEX].to(torch.int64)
    ok = (idx >= 0) & (idx < global_ids.shape[0])
    out = records.clone()
    if global_ids.shape[0] == 0:
        out[:, COL_INDEX] = -1
        return out
    gid = torch.where(ok, global_ids[idx.clamp(0, global_ids.shape[0] - 1)], torch.full_like(idx, -1))
    out[:, COL_INDEX] = gid.to(torch.int32)
    return out
