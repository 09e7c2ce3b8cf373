"""The synthetic database of bench.py / the -m gpu config tests (hh-suite_amd/pyhhv/synth_stream.py): the random streams
SURVEY.md 8(d) prescribes - splitmix64 -> xoshiro256**, seed 0x5EED0000 + global template id, u = (x >> 40) * 2^-24 -
pinned to the generators' published first outputs, the torch version (what runs on the GPU) to the numpy restatement,
and the property the multi-GPU runs rely on: a template's columns depend on its global id alone.  Plus bench.py's
global plan (ONE database, hhv_shard_plan, every rank the same answer).  CPU only."""
import os
import sys
import types

import numpy as np
import torch

from pyhhv import capi, synth, synth_stream as ss

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_known_answers_of_both_generators():
    # splitmix64, seed 1234567 (Vigna's reference implementation; the vector every port quotes)
    st = ss.splitmix64_states(np.array([1234567], dtype=np.uint64))
    assert [int(x) for x in st[:, 0]] == [6457827717110365317, 3203168211198807973, 9817491932198370423, 4593380528125082431]
    # xoshiro256** from the state {1, 2, 3, 4}
    s = np.array([[1], [2], [3], [4]], dtype=np.uint64)
    assert [int(ss.xoshiro_next_np(s)[0]) for _ in range(4)] == [11520, 0, 1509978240, 1215971899390074240]


def test_torch_streams_equal_numpy_streams():
    gids = np.array([0, 1, 2, 77, 99999, 999999])
    a = ss.uniforms_np(gids, 300)
    b = ss.uniforms_torch(torch, torch.device("cpu"), gids, 300).numpy()
    assert np.array_equal(a, b)
    assert a.min() >= 0.0 and a.max() < 1.0 and 0.4 < a.mean() < 0.6
    assert len(np.unique(a[:, 0])) == len(gids)            # different seeds, different streams


def test_template_columns_depend_on_the_global_id_only():
    n = 40
    gids = 1000 + np.arange(n)
    Ls = synth.zipf_lengths(0x21F, n)
    dev = torch.device("cpu")
    rec, off, L = ss.gen_stream(torch, dev, gids, Ls, synth.PB)
    perm = np.random.RandomState(3).permutation(n)
    rec2, off2, L2 = ss.gen_stream(torch, dev, gids[perm], Ls[perm], synth.PB)
    a, b = rec.numpy(), rec2.numpy()
    for k2, k in enumerate(perm):
        x = a[off[k] + 1: off[k] + 1 + L[k]]
        y = b[off2[k2] + 1: off2[k2] + 1 + L2[k2]]
        assert np.array_equal(x.view(np.int32), y.view(np.int32)), k
        hx, hy = a[off[k]].view(np.int32), b[off2[k2]].view(np.int32)
        assert hx[27] == hy[27] == -2 ** 31 and hx[0] == k and hy[0] == k2 and hx[1] == hy[1] == L[k]
    # stream layout: meta = column index, last-column flag, terminal header
    m = a.view(np.int32)
    assert m[off[0] + 1, 27] == 1 and m[off[0] + L[0], 27] == (int(L[0]) | 0x40000000)
    assert m[off[-1], 27] == -2 ** 31 and m[off[-1], 0] == -1
    # the records are what the product's packer produces from the unpacked profiles
    tps, ttrs = ss.unpack_templates(a[: off[-1]], off, L, n)
    for k in (0, 7, n - 1):
        packed = capi.pack_profile(tps[k], ttrs[k], index=k)
        assert np.array_equal(packed[1:, :27].view(np.int32), a[off[k] + 1: off[k] + 1 + L[k], :27].view(np.int32))


def test_unpack_fast_path_equals_general_path():
    n, Lt = 12, 17
    rec, off, L = ss.gen_stream(torch, torch.device("cpu"), np.arange(n), np.full(n, Lt), synth.PB)
    h = rec[: off[-1]].numpy()
    fast = ss.unpack_templates(h, off, L, n)
    for k in range(n):
        p, tr = ss.unpack_one(h[off[k]: off[k + 1]], Lt)
        assert np.array_equal(fast[0][k], p) and np.array_equal(fast[1][k], tr)


def _args(lengths, lt=300):
    return types.SimpleNamespace(lengths=lengths, lt=lt)


def test_bench_global_plan_is_one_database_cut_by_hhv_shard_plan():
    import bench
    for lengths in ("fixed", "zipf"):
        Lg, ids, records = bench.global_plan(_args(lengths, 120), capi, synth, 1500, 4)
        assert Lg.shape == (6000,)
        allids = np.concatenate(ids)
        assert np.array_equal(np.sort(allids), np.arange(6000))                    # disjoint cover of the global ids
        owner = capi.shard_plan(Lg, 4)
        for r, g in enumerate(ids):
            assert np.all(owner[g] == r)
            assert np.all(np.diff(Lg[g]) <= 0)                                     # longest first inside a shard
            assert records[r] == int(np.sum(Lg[g].astype(np.int64) + 1))
        assert records.max() / records.mean() < 1.02
        if lengths == "fixed":
            assert [len(g) for g in ids] == [1500] * 4 and np.array_equal(ids[1], 1500 + np.arange(1500))
        else:
            assert Lg.min() >= 50 and Lg.max() <= 1000
            # the same length vector whatever the number of shards: prefix property of the one seed
            Lg8, _, _ = bench.global_plan(_args(lengths, 120), capi, synth, 750, 8)
            assert np.array_equal(Lg8, Lg)


def test_bench_defaults_are_baseline_configs():
    import bench
    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse()
        assert a.gpus == 1 and a.templates is None and a.lq == 300 and a.lt == 300 and a.topk == 500
    finally:
        sys.argv = old
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "100000 if args.gpus == 1 else 125000" in src      # 8 x 125 000 = configs[3]'s 1 M templates


def test_segment_plan_of_the_work_queue():
    """hhv_segment_plan (host logic of the DP kernel's work queue, hhv_api.cpp plan_segments): whole templates, every record
    exactly once, >= 128 records per segment unless the whole stream is shorter, drawn longest first (stable), terminal entry."""
    from pyhhv import capi, synth
    rng = np.random.default_rng(5)
    cases = [np.full(1000, 300), np.full(500, 127), np.full(7, 20), np.array([20]), np.array([60, 30]), np.array([200, 200, 30]),
             synth.zipf_lengths(0x21F, 20000), rng.integers(1, 40, 3000), np.array([1000, 1, 1, 1]), np.zeros(0, dtype=np.int32)]
    for L in cases:
        L = np.asarray(L, dtype=np.int32)
        n = L.shape[0]
        n_seg, seg = capi.segment_plan(L)
        off = np.concatenate([[0], np.cumsum(L.astype(np.int64) + 1)])
        total = int(off[-1])
        assert seg.shape == (n_seg + 1, 2) and tuple(seg[n_seg]) == (total, total + 1)
        if n == 0:
            assert n_seg == 0
            continue
        body = seg[:n_seg]
        length = body[:, 1] - body[:, 0]
        assert np.all(length[:-1] >= length[1:])                                  # longest first ...
        same = length[:-1] == length[1:]
        assert np.all(body[:-1, 0][same] < body[1:, 0][same])                     # ... stable: equal lengths in stream order
        by_pos = body[np.argsort(body[:, 0])]
        assert by_pos[0, 0] == 0 and by_pos[-1, 1] == total and np.array_equal(by_pos[1:, 0], by_pos[:-1, 1])   # a partition
        assert np.all(np.isin(by_pos[:, 0], off))                                  # cut at template boundaries only
        assert n_seg == 1 or np.all(length >= 128)
        assert n_seg <= n
        # closed as soon as it holds 128 records: without its last template a segment is shorter (the last one excepted)
        ends = np.searchsorted(off, by_pos[:-1, 1])
        assert np.all(off[ends - 1] - by_pos[:-1, 0] < 128)
    assert capi.load().hhv_segment_plan(2, np.array([5, 0], dtype=np.int32).ctypes.data_as(capi.c_int_p), seg.ctypes.data_as(capi.C.POINTER(capi.C.c_int64)),
                                        capi.C.byref(capi.C.c_int32(0))) != 0
