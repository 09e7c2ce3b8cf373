"""Multi-rank path on CPU: world_size-2 and -3 gloo.  Each rank "aligns" its shard with the oracle (standing
in for its GPU), builds the K-record buffer exactly like hhv_topk lays it out, and the exchange +
merge of pyhhv.shard must give every rank the global top-K a single process computes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pyhhv import shard, synth
from pyoracle import Oracle, make_params


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_db(n, seed=3):
    rng = np.random.default_rng(seed)
    qf, qtr = synth.make_query(77, 60)
    tps, ttrs = [], []
    for k in range(n):
        L = int(rng.integers(20, 90))
        p, tr = synth.make_homolog(300 + k, qf, L=L) if k % 3 == 0 else synth.make_template(300 + k, L)
        tps.append(p)
        ttrs.append(tr)
    return qf, qtr, tps, ttrs


def local_records(o, par, qf, qtr, tps, ttrs, ids, K):
    rec = np.full((K, shard.REC_I32), -1, dtype=np.int32)
    rows = []
    for loc, g in enumerate(ids):
        a = o.align(par, qf, qtr, tps[g], ttrs[g], want_bt=False)
        rows.append((float(a.score), loc, a.i2, a.j2))
    rows.sort(key=lambda r: (-r[0], r[1]))
    for t, (s, loc, i2, j2) in enumerate(rows[:K]):
        bits = np.float32(s).view(np.int32)
        rec[t] = [bits, bits, 0, loc, 0, 0, i2, j2, 0, 0]
    return rec


def worker(rank, world, port, n, K, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = Oracle()
    par = make_params(local=1)
    qf, qtr, tps, ttrs = make_db(n)
    Ls = [p.shape[0] - 1 for p in tps]
    parts = shard.shard_templates(Ls, world)
    ids = parts[rank]
    rec = torch.from_numpy(local_records(o, par, qf, qtr, tps, ttrs, ids, K))
    rec = shard.to_global_ids(torch, rec, torch.from_numpy(ids.astype(np.int64)))
    merged = shard.exchange_and_merge(torch, dist, rec, K)
    np.save(os.path.join(out_dir, "merged_%d.npy" % rank), merged.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,K", [(2, 7), (3, 20)])    # (3, 20): a shard holds fewer templates than K - padded records
def test_multi_rank_topk_merge(tmp_path, world, K):
    n = 40
    port = free_port()
    mp.spawn(worker, args=(world, port, n, K, str(tmp_path)), nprocs=world, join=True)
    m0 = np.load(tmp_path / "merged_0.npy")
    for r in range(1, world):
        assert np.array_equal(m0, np.load(tmp_path / ("merged_%d.npy" % r))), "ranks disagree on the merged hit list"
    o = Oracle()
    par = make_params(local=1)
    qf, qtr, tps, ttrs = make_db(n)
    scores = [float(o.align(par, qf, qtr, tps[g], ttrs[g], want_bt=False).score) for g in range(n)]
    want = sorted(range(n), key=lambda g: (-scores[g], g))[:K]
    assert list(m0[:, shard.COL_INDEX]) == want
    assert np.array_equal(m0[:, 0].view(np.float32), np.array([scores[g] for g in want], dtype=np.float32))


def test_shard_partition_properties():
    rng = np.random.default_rng(0)
    Ls = synth.zipf_lengths(5, 5000)
    for world in (1, 2, 4, 8):
        parts = shard.shard_templates(Ls, world)
        allids = np.concatenate(parts)
        assert sorted(allids.tolist()) == list(range(5000))
        loads = np.array([(Ls[p] + 1).sum() for p in parts], dtype=np.float64)
        assert loads.max() / loads.mean() < 1.05
    eq = shard.shard_templates(np.full(1000, 300), 8)
    assert [len(p) for p in eq] == [125] * 8 and np.array_equal(np.concatenate(eq), np.arange(1000))


def test_merge_tie_break():
    rec = torch.tensor([[np.float32(1.5).view(np.int32), 0, 0, 9, 0, 0, 0, 0, 0, 0],
                        [np.float32(2.5).view(np.int32), 0, 0, 4, 0, 0, 0, 0, 0, 0],
                        [np.float32(1.5).view(np.int32), 0, 0, 3, 0, 0, 0, 0, 0, 0],
                        [-1] * 10], dtype=torch.int32)
    m = shard.merge_records(torch, rec, 3)
    assert list(m[:, shard.COL_INDEX]) == [4, 3, 9]


def test_c_abi_shard_plan_equals_python_partition():
    """hhv_shard_plan (the partition the C++ hosts use) == pyhhv.shard.shard_templates, ragged and equal lengths."""
    from pyhhv import capi
    for Ls in (synth.zipf_lengths(9, 3000), np.full(1000, 300, dtype=np.int32), synth.zipf_lengths(2, 37), np.array([5], dtype=np.int32)):
        for world in (1, 2, 3, 8):
            plan = capi.shard_plan(Ls, world)
            parts = shard.shard_templates(Ls, world)
            for r, ids in enumerate(parts):
                assert np.all(plan[ids] == r)
            assert sum(len(p) for p in parts) == len(Ls)
