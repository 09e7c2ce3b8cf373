"""Side entries of bench.py for the hot-path rows the headline configuration does not exercise (never part of `value`):

  ss_modes      SURVEY 8a A5: the ...AndSS kernels (par.ssm = 2 is the reference's default, src/hhdecl.cpp:82; they run whenever the
                templates of a batch carry secondary-structure records, src/hhviterbi.cpp:175) - PRED_PRED and DSSP_PRED, score-only
                and with backtrace + hits, on the headline set with secondary-structure codes added to the stream's meta words, and
                a two-strip query (Lq 431); each mode checked against the reference (oracle/_ref) on a sample
  multi_strip   queries of two and more strips (Lq 431, 512, 1000): pair / chain launches against one launch per strip in the
                same process (hhv_set_launch_policy)
  masked_round  SURVEY 8a A4: the second alternative-alignment round - masks built on the device from the first round's paths
                (hhv_set_celloff_paths = Viterbi::ExcludeAlignment) and the masked DP (AlignWithCellOff) over 10 k templates

Every entry = {cells_per_s, ms_per_step, dp_kernel_ms, ...}; a step is hhv_set_query + hhv_align_async (+ hhv_hits) + hhv_topk."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "hh-suite_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

SS_SEED = 0x55AA
META_PRED_SHIFT, META_DSSP_SHIFT = 16, 22


def ss_tables(seed=SS_SEED):
    """Score tables of the shape the reference reads (S73[8][4][11], S33[4][11][4][11], S37[4][11][8], src/hhdecl.h); values are
    log-odds-like random numbers - the kernels do not look at them, the cross-check runs on the same tables."""
    rng = np.random.default_rng(seed)
    return (rng.normal(0, 1, (8, 4, 11)).astype(np.float32), rng.normal(0, 1, (4, 11, 4, 11)).astype(np.float32),
            rng.normal(0, 1, (4, 11, 8)).astype(np.float32))


def query_ss(Lq, seed=SS_SEED + 1):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 4, Lq + 1).astype(np.int8), rng.integers(0, 11, Lq + 1).astype(np.int8),
            rng.integers(0, 8, Lq + 1).astype(np.int8))


def add_ss_codes(torch, rec, nrec):
    """Secondary-structure codes for every column record of a packed stream, written into its meta word the way
    hhv_upload_templates_ss packs them (viterbi_lane.h META_PRED_SHIFT / META_DSSP_SHIFT; src/hhhmmsimd.cpp:132-135:
    pred_index = ss_pred * 11 + ss_conf, dssp_index = ss_dssp).  A hash of the record number: the same set on every run."""
    meta = rec.view(torch.int32)[:nrec, 27]
    pos = torch.arange(nrec, dtype=torch.int64, device=rec.device)
    h = (pos * 2654435761 + 12345) & 0x7FFFFFFF
    h = (h ^ (h >> 13)) * 1274126177 & 0x7FFFFFFF
    pred = (h >> 3) % 4
    conf = (h >> 7) % 11
    dssp = (h >> 17) % 8
    code = (((pred * 11 + conf) << META_PRED_SHIFT) | (dssp << META_DSSP_SHIFT)).to(torch.int32)
    col = meta >= 0   # (headers keep their meta word)
    meta[col] = (meta[col] & 0x4000FFFF) | code[col]


def template_ss_of(meta_host, L):
    """(ss_pred, ss_conf, ss_dssp) [L+1] of one template from the meta words of its L+1 records (header first)"""
    m = np.asarray(meta_host, dtype=np.int64)
    pi = (m >> META_PRED_SHIFT) & 0x3F
    ds = (m >> META_DSSP_SHIFT) & 0x7
    pi[0] = 0
    ds[0] = 0
    return (pi // 11).astype(np.int8), (pi % 11).astype(np.int8), ds.astype(np.int8)


def _time_steps(ctx, ts, qf, qtr, K, bt, reps, warm=2, q_ss=None):
    """a step = a search: the query (and its secondary structure: hhv_set_query forgets the previous query's) goes to the device"""
    ms = []
    for it in range(warm + reps):
        if it == warm:
            ctx.sync()
            t0 = time.perf_counter()
        ctx.set_query(qf, qtr)
        if q_ss is not None:
            ctx.set_query_ss(*q_ss)
        ctx.align_async(ts, backtrace=bt)
        if bt:
            ctx.hits(ts, fetch=False)
        ctx.topk(ts, K, fetch=False, raw=not bt)
        if it >= warm:
            ms.append(ctx.last_kernel_ms())
    ctx.sync()
    sec = (time.perf_counter() - t0) / reps
    return {"cells_per_s": ts.cells() / sec, "ms_per_step": sec * 1e3, "dp_kernel_ms": float(np.mean(ms)),
            "dp_kernel_ms_min": float(np.min(ms))}


def _check_ss(ctx, ts, rec, rec_off, Ls, qf, qtr, q_ss, tables, mode, local, m):
    """GPU (backtrace launch + hits) against the reference's ...AndSS build on the first m templates: end points, scores,
    Hit.score, score_ss, path lengths"""
    import pyoracle
    from pyhhv import synth_stream
    S73, S33, S37 = tables
    par = pyoracle.make_params(local=local, ss_mode=2)
    use_ref = pyoracle.have_ref()
    host = rec[: int(rec_off[m])].cpu().numpy()
    tps, ttrs = synth_stream.unpack_templates(host, rec_off, Ls, m)
    metas = host.view(np.int32)[:, 27]
    t_sss = [template_ss_of(metas[int(rec_off[k]): int(rec_off[k + 1])], int(Ls[k])) for k in range(m)]
    ss = pyoracle.SSInfo(mode, q_ss[0], q_ss[1], q_ss[2], S73, S33, S37)
    ctx.set_query(qf, qtr)
    ctx.set_query_ss(*q_ss)
    res = ctx.align(ts, backtrace=True)
    hits = ctx.hits(ts)
    bad = 0
    maxd = 0.0
    if use_ref:
        eng = pyoracle.Ref()
        V = eng.V
        for b in range(0, m, V):
            outs = eng.align_batch(par, qf, qtr, tps[b:b + V], ttrs[b:b + V], replicate=False, ss=ss, t_sss=t_sss[b:b + V], want_bt=False,
                                   want_path=True)
            for e, a in enumerate(outs):
                k = b + e
                ok = ((a.i2, a.j2) == (int(res["i2"][k]), int(res["j2"][k])) and np.float32(a.score) == res["score"][k] and
                      np.float32(a.hit_score) == hits["score"][k] and np.float32(a.score_ss) == hits["score_ss"][k] and a.nsteps == int(hits["nsteps"][k]))
                bad += 0 if ok else 1
                maxd = max(maxd, abs(float(a.score) - float(res["score"][k])))
    else:
        eng = pyoracle.Oracle()
        m = min(m, 64)
        for k in range(m):
            a = eng.align(par, qf, qtr, tps[k], ttrs[k], ss=ss, t_ss=t_sss[k], want_path=True)
            ok = ((a.i2, a.j2) == (int(res["i2"][k]), int(res["j2"][k])) and np.float32(a.score) == res["score"][k] and
                  np.float32(a.hit_score) == hits["score"][k] and np.float32(a.score_ss) == hits["score_ss"][k] and a.nsteps == int(hits["nsteps"][k]))
            bad += 0 if ok else 1
            maxd = max(maxd, abs(float(a.score) - float(res["score"][k])))
    # the score-only kernel against the backtrace kernel on the WHOLE set
    r1 = ctx.align(ts)
    same = bool(np.array_equal(r1.view(np.uint8), res.view(np.uint8)))
    return {"templates_checked": m, "cpu_kind": "reference" if use_ref else "port", "mismatches": bad, "max_abs_score_diff": maxd,
            "score_only_equals_backtrace_launch_on_all": same}


def ss_modes(torch, capi, dev_index, rec, rec_off, Ls, qf, qtr, K, local=0, reps=8, lq2=431, n2=50000, check=512):
    """rec: the resident stream of the headline set (NOT modified: the codes go into a copy)."""
    from pyhhv import synth, synth_stream
    out = {"what": "...AndSS kernels (ssm 2, src/hhviterbi.cpp:175): headline set + secondary-structure codes in the meta words; "
                   "PRED_PRED = S33[q_pred][q_conf][t_pred][t_conf] (the only mode ViterbiRunner ever selects, src/hhviterbirunner.cpp:14-22), "
                   "DSSP_PRED = S73[q_dssp][t_pred][t_conf]"}
    n = int(Ls.shape[0])
    nrec = int(rec_off[n])
    rs = rec.clone()
    add_ss_codes(torch, rs, nrec)
    tables = ss_tables()
    Lq = qf.shape[0] - 1
    try:
        c = capi.Context(local=local, device=dev_index, ss_mode=2)
        c.set_query(qf, qtr)
        c.set_ss_tables(*tables)
        q_ss = query_ss(Lq)
        c.set_query_ss(*q_ss)
        ts = c.adopt_device_stream(Ls, rs.data_ptr())
        c.set_ss_mode(0)
        out["no_ss_same_stream"] = {"score_only": _time_steps(c, ts, qf, qtr, K, False, reps), "backtrace_hits": _time_steps(c, ts, qf, qtr, K, True, max(3, reps // 2))}
        for mode, name in ((4, "PRED_PRED"), (2, "DSSP_PRED")):
            c.set_ss_mode(mode)
            e = {"score_only": _time_steps(c, ts, qf, qtr, K, False, reps, q_ss=q_ss),
                 "backtrace_hits": _time_steps(c, ts, qf, qtr, K, True, max(3, reps // 2), q_ss=q_ss)}
            try:
                e["gpu_matches_cpu_on_sample"] = _check_ss(c, ts, rs, rec_off, Ls, qf, qtr, q_ss, tables, mode, local, min(check, n))
            except Exception as ex:
                e["gpu_matches_cpu_on_sample"] = {"error": repr(ex)}
            out[name] = e
        ts.free()
        # two strips
        if n >= n2:
            q2f, q2tr = synth_stream.query_np(lq2, synth.PB)
            c.set_query(q2f, q2tr)
            q2_ss = query_ss(lq2)
            c.set_query_ss(*q2_ss)
            ts2 = c.adopt_device_stream(Ls[:n2], rs.data_ptr())
            for mode, name in ((0, "no_ss"), (4, "PRED_PRED")):
                c.set_ss_mode(mode)
                out["Lq%d_%dk_%s" % (lq2, n2 // 1000, name)] = {"score_only": _time_steps(c, ts2, q2f, q2tr, K, False, reps, q_ss=q2_ss),
                                                               "backtrace_hits": _time_steps(c, ts2, q2f, q2tr, K, True, max(3, reps // 2), q_ss=q2_ss)}
            c.set_ss_mode(4)
            try:
                out["Lq%d_%dk_PRED_PRED" % (lq2, n2 // 1000)]["gpu_matches_cpu_on_sample"] = _check_ss(
                    c, ts2, rs, rec_off, Ls, q2f, q2tr, q2_ss, tables, 4, local, min(check // 2, n2))
            except Exception as ex:
                out["Lq%d_%dk_PRED_PRED" % (lq2, n2 // 1000)]["gpu_matches_cpu_on_sample"] = {"error": repr(ex)}
            ts2.free()
        c.close()
    except Exception as ex:  # the headline line must not depend on the side measurements
        out["error"] = repr(ex)
    del rs
    return out


def multi_strip(torch, capi, dev_index, rec, rec_off, Ls, K, local=0, reps=8):
    """Lq 431 (strips 4 + 3), 512 (4 + 4), 1000 (four strips) over the first templates of the resident stream (the stream's
    own lengths), score-only and with backtrace + hits; `one_launch_per_strip` = the same search with pair / chain launches
    switched off (hhv_set_launch_policy pair_mode 0) in the same process."""
    from pyhhv import synth, synth_stream
    out = {}
    n = int(Ls.shape[0])
    try:
        c = capi.Context(local=local, device=dev_index)
        for lq, m in ((431, 50000), (512, 50000), (1000, 30000)):
            m = min(m, n)
            qf, qtr = synth_stream.query_np(lq, synth.PB)
            c.set_query(qf, qtr)
            ts = c.adopt_device_stream(Ls[:m], rec.data_ptr())
            e = {"templates": m}
            for bt, name in ((False, "score_only"), (True, "backtrace_hits")):
                r = max(3, reps // 2) if bt else reps
                c.set_launch_policy(pair_mode=-1)
                e[name] = _time_steps(c, ts, qf, qtr, K, bt, r)
                a = c.align(ts, backtrace=bt)
                c.set_launch_policy(pair_mode=0)
                e[name]["one_launch_per_strip"] = _time_steps(c, ts, qf, qtr, K, bt, r)
                b = c.align(ts, backtrace=bt)
                e[name]["results_identical"] = bool(np.array_equal(a.view(np.uint8), b.view(np.uint8)))
                c.set_launch_policy(pair_mode=-1)
            out["Lq%d" % lq] = e
            ts.free()
        c.close()
    except Exception as ex:
        out["error"] = repr(ex)
    return out


def masked_round(torch, capi, dev_index, rec, rec_off, Ls, qf, qtr, K, local=0, n10=10000, reps=5, check=256):
    """Round 1 (backtrace + hits) over n10 templates, then the second alternative-alignment round of every template:
    hhv_set_celloff_paths (the reference's Viterbi::ExcludeAlignment for every path, src/hhviterbi.cpp:61-77, on the device)
    + the masked DP (AlignWithCellOff, src/hhviterbialgorithm.cpp:373-392) + hits; checked against the reference on a sample."""
    import pyoracle
    from pyhhv import synth_stream
    out = {"templates": n10}
    try:
        c = capi.Context(local=local, device=dev_index)
        c.set_query(qf, qtr)
        ts = c.adopt_device_stream(Ls[:n10], rec.data_ptr())
        c.align(ts, backtrace=True)
        hits1 = c.hits(ts)
        off, pi, pj, pst, pS = c.hit_path_pool(ts)
        paths = [(k, int(hits1["nsteps"][k]), pi[off[k]: off[k] + hits1["nsteps"][k] + 1], pj[off[k]: off[k] + hits1["nsteps"][k] + 1]) for k in range(n10)]
        packed = c.pack_celloff_paths(paths)   # (the host arrays hhv_set_celloff_paths takes: built once, outside the timing)
        t_mask, t_round, kms = [], [], []
        for it in range(reps + 1):
            c.sync()
            t0 = time.perf_counter()
            c.set_celloff_paths_packed(ts, packed)
            c.sync()
            t1 = time.perf_counter()
            c.align_async(ts, celloff=True)
            c.hits(ts, fetch=False)
            c.topk(ts, K, fetch=False)
            c.sync()
            t2 = time.perf_counter()
            if it:
                t_mask.append(t1 - t0)
                t_round.append(t2 - t1)
                kms.append(c.last_kernel_ms())
        cells = ts.cells()
        out.update({"masked_dp_cells_per_s": cells / (float(np.mean(kms)) * 1e-3), "dp_kernel_ms": float(np.mean(kms)),
                    "round_ms_dp_hits_topk": float(np.mean(t_round)) * 1e3, "cells_per_s_round": cells / float(np.mean(t_round)),
                    "mask_build_ms_hhv_set_celloff_paths": float(np.mean(t_mask)) * 1e3,
                    "path_steps_handed_over": int(packed[1][-1])})
        try:
            use_ref = pyoracle.have_ref()
            m = min(check if use_ref else 32, n10)
            c.set_celloff_paths_packed(ts, packed)
            res2 = c.align(ts, celloff=True)
            hits2 = c.hits(ts)
            host = rec[: int(rec_off[m])].cpu().numpy()
            tps, ttrs = synth_stream.unpack_templates(host, rec_off, Ls, m)
            par = pyoracle.make_params(local=local)
            orc = pyoracle.Oracle()
            Lq = qf.shape[0] - 1
            bad = 0
            if use_ref:
                eng = pyoracle.Ref()
                V = eng.V
                for b in range(0, m, V):
                    masks = [orc.exclude_alignment(Lq, int(Ls[k]), paths[k][2], paths[k][3], paths[k][1]) for k in range(b, min(b + V, m))]
                    outs = eng.align_batch(par, qf, qtr, tps[b:b + V], ttrs[b:b + V], replicate=False, celloffs=masks, want_bt=False, want_path=True)
                    for e, a in enumerate(outs):
                        k = b + e
                        ok = ((a.i2, a.j2) == (int(res2["i2"][k]), int(res2["j2"][k])) and np.float32(a.score) == res2["score"][k] and
                              np.float32(a.hit_score) == hits2["score"][k] and a.nsteps == int(hits2["nsteps"][k]))
                        bad += 0 if ok else 1
            else:
                for k in range(m):
                    mask = orc.exclude_alignment(Lq, int(Ls[k]), paths[k][2], paths[k][3], paths[k][1])
                    a = orc.align(par, qf, qtr, tps[k], ttrs[k], celloff=mask, want_path=True)
                    ok = ((a.i2, a.j2) == (int(res2["i2"][k]), int(res2["j2"][k])) and np.float32(a.score) == res2["score"][k] and
                          np.float32(a.hit_score) == hits2["score"][k] and a.nsteps == int(hits2["nsteps"][k]))
                    bad += 0 if ok else 1
            out["gpu_matches_cpu_on_sample"] = {"templates_checked": m, "cpu_kind": "reference" if use_ref else "port", "mismatches": bad}
        except Exception as ex:
            out["gpu_matches_cpu_on_sample"] = {"error": repr(ex)}
        ts.free()
        c.close()
    except Exception as ex:
        out["error"] = repr(ex)
    return out
