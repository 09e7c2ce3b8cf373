#!/usr/bin/env python3
"""Randomised parity soak on a GPU box: the HIP kernels against the oracle (test infrastructure) on many small random
cases per kernel family, with adversarial inputs the fixed tests do not have (exact ties, -100000 transitions, saturated
prefilter profiles, random cell-off masks, length 1).  usage: python tools/soak.py [seconds_per_family] -> JSON line."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in ("hh-suite_amd", "oracle", "tests"):
    sys.path.insert(0, os.path.join(ROOT, d))
from pyhhv import capi, synth  # noqa: E402
import pyoracle as po  # noqa: E402


def quantize(a, rng, step):
    """Coarse grid -> many exact ties between DP candidates."""
    return (np.round(a / step) * step).astype(np.float32)


def viterbi_family(orc, rng, budget, with_ss=False):
    """with_ss: the ...AndSS kernels (par.ssm = 2, a random table mode, random secondary-structure codes for query and templates,
    random score tables) - the same cases otherwise: ties, dead transitions, masked rounds (AlignWithCellOffAndSS), thousands of
    1-3-column templates back to back, queries of one to four strips and the short-query arrays"""
    t_end, cases, bad = time.time() + budget, 0, 0
    while time.time() < t_end:
        Lq = int(rng.choice([1, 2, 5, 63, 64, 65, 100, 320, 321, 400, 512, 600, 700, 1000]))   # (321 ..: pair kernels, 641 ..: chains of them)
        local = int(rng.integers(0, 2))
        par = po.make_params(local=local, egq=float(rng.choice([0.0, 0.2])), egt=float(rng.choice([0.0, 0.1])),
                             shift=float(rng.choice([-0.03, 0.0, 0.25])), ss_mode=2 if with_ss else 0)
        qp, qtr = synth.make_query(int(rng.integers(1 << 30)), Lq)
        n = int(rng.integers(1, 12))
        tps, ttrs, masks, t_sss = [], [], [], []
        ss = None
        if with_ss:
            par["ssw"] = float(rng.choice([0.11, 1.0, 0.0]))
            tables = (rng.normal(0, 1, (8, 4, 11)).astype(np.float32), rng.normal(0, 1, (4, 11, 4, 11)).astype(np.float32),
                      rng.normal(0, 1, (4, 11, 8)).astype(np.float32))
            if rng.random() < 0.3:   # ties between the candidates of a cell survive the addition of a coarse table
                tables = tuple(quantize(t, rng, 0.5) for t in tables)
            q_ss = (rng.integers(0, 4, Lq + 1), rng.integers(0, 11, Lq + 1), rng.integers(0, 8, Lq + 1))
            ss = po.SSInfo(int(rng.choice([4, 4, 2, 1])), *q_ss, *tables)
        for k in range(n):
            Lt = int(rng.choice([1, 2, 3, 31, 64, 130, 257]))
            tp, ttr = (synth.make_homolog(int(rng.integers(1 << 30)), qp, L=Lt) if rng.random() < 0.5 and Lq > 4 else
                       synth.make_template(int(rng.integers(1 << 30)), Lt))
            if rng.random() < 0.5:       # ties
                ttr = quantize(ttr, rng, 0.5)
                ttr[ttr < -1000] = -100000.0
            if rng.random() < 0.2:       # impossible transitions in the middle
                ttr[rng.integers(0, Lt + 1), rng.integers(0, 7)] = -100000.0
            tps.append(tp)
            ttrs.append(ttr)
            t_sss.append((rng.integers(0, 4, Lt + 1), rng.integers(0, 11, Lt + 1), rng.integers(0, 8, Lt + 1)))
            masks.append((rng.random((Lq + 1, Lt + 1)) < rng.choice([0.0, 0.05, 0.5])).astype(np.uint8) if rng.random() < 0.4 else None)
        if rng.random() < 0.5:
            qtr = quantize(qtr, rng, 0.5)
            qtr[qtr < -1000] = -100000.0
        # a third of the cases: thousands of copies of the pool in random order, so that every resident wavefront walks several
        # templates - most of them 1-3 columns long - back to back (headers in consecutive steps: the finalized-best hand-off of
        # the 64-lane arrays, the column-0 reset behind a last column); the pool alone is one template per wavefront
        many = rng.random() < 0.35
        idx = rng.integers(0, n, int(rng.integers(2500, 6000))) if many else np.arange(n)
        if many and rng.random() < 0.7:
            short = [k for k in range(n) if tps[k].shape[0] - 1 <= 3]
            if short:
                idx = np.where(rng.random(idx.shape[0]) < 0.6, rng.choice(short, idx.shape[0]), idx)
        c = capi.Context(local=local, egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"], ssw=par["ssw"], ss_mode=par["ss_mode"])
        c.set_query(qp, qtr)
        if with_ss:
            c.set_ss_tables(ss.S73, ss.S33, ss.S37)
            c.set_query_ss(ss.q_pred, ss.q_conf, ss.q_dssp)
            c.set_ss_mode(ss.mode)
        ts = c.upload([tps[k] for k in idx], [ttrs[k] for k in idx], [t_sss[k] for k in idx] if with_ss else None)
        use_mask = any(m is not None for m in masks)
        if use_mask:
            for e, k in enumerate(idx):
                if many and e >= 64 and masks[k] is None:
                    continue   # (a set that never saw a compare-bit launch starts with all cells on)
                m = masks[k]
                c.set_celloff(ts, e, m if m is not None else np.zeros((Lq + 1, tps[k].shape[0]), np.uint8))
        # one case in five score-only (other kernels; two equal strips run as a pair only then): end points and scores
        with_bt = use_mask or rng.random() < 0.8
        res = c.align(ts, backtrace=with_bt, celloff=use_mask)
        hits = c.hits(ts) if with_bt else None
        want = [orc.align(par, qp, qtr, tps[k], ttrs[k], celloff=masks[k] if use_mask else None, ss=ss, t_ss=t_sss[k] if with_ss else None,
                          want_path=True) for k in range(n)]
        bt_sample = set(range(len(idx))) if not many else set(int(e) for e in rng.integers(0, len(idx), 12))
        for e, k in enumerate(idx):
            a = want[k]
            ok = (a.i2, a.j2) == (int(res["i2"][e]), int(res["j2"][e])) and np.float32(a.score).tobytes() == np.float32(res["score"][e]).tobytes()
            if with_bt and e in bt_sample:
                ok = ok and np.array_equal(c.backtrace_matrix(ts, e)[1:, 1:] & 0x7F, a.bt[1:, 1:] & 0x7F)
            if with_bt:
                ok = ok and int(hits["nsteps"][e]) == a.nsteps and np.float32(hits["score"][e]).tobytes() == np.float32(a.hit_score).tobytes()
                # (a template whose every cell is masked ends at (0, 0): the reference then evaluates ScoreSS at index 0 of the
                # secondary-structure arrays, which HMM::Read never writes (src/hhhmm.cpp:324-455 fill from 1) - undefined there,
                # 0 here; Hit.score is -FLT_MAX either way)
                if with_ss and a.i2 >= 1:
                    ok = ok and np.float32(hits["score_ss"][e]) == np.float32(a.score_ss)
            cases += 1
            bad += int(not ok)
            if not ok and bad <= 6:
                print("MISMATCH viterbi%s: Lq %d local %d Lt %d copies %d entry %d mask %s bt %s par %s%s | oracle (%d, %d) %r gpu (%d, %d) %r%s" % (
                    " ss" if with_ss else "", Lq, local, tps[k].shape[0] - 1, len(idx), e, masks[k] is not None and use_mask, with_bt, par,
                    " mode %d" % ss.mode if with_ss else "", a.i2, a.j2, float(a.score), int(res["i2"][e]), int(res["j2"][e]), float(res["score"][e]),
                    " hits: nsteps %d / %d score %r / %r ss %r / %r" % (a.nsteps, int(hits["nsteps"][e]), float(a.hit_score), float(hits["score"][e]),
                                                                      float(a.score_ss), float(hits["score_ss"][e])) if with_bt else ""), file=sys.stderr)
        ts.free()
        c.close()
    return {"cases": cases, "mismatches": bad}


def long_family(orc, rng, budget):
    """Lengths beyond the 2100 columns the fixed tests and viterbi_family stop at, up to the reference's maxres (20 001, src/hhdecl.cpp:11)
    and the library's own limits in one dimension: many passes (carry rows, the running best across passes), 16-bit column indices,
    path pools of tens of thousands of steps; ties, -0.0f and dead transitions, a masked second round on the first template."""
    t_end, cases, bad = time.time() + budget, 0, 0
    while time.time() < t_end:
        Lq = int(rng.choice([2101, 3000, 4097, 7777, 12000, 20001, 321, 640]))
        lt_pool = [2101, 2500, 5000, 9000, 20001, 40000, 65535] if Lq <= 4097 else [77, 1500, 2101, 6000, 20001]
        local = int(rng.integers(0, 2))
        par = po.make_params(local=local, egq=float(rng.choice([0.0, 0.2])), egt=float(rng.choice([0.0, 0.1])), ss_mode=0)
        qp, qtr = synth.make_query(int(rng.integers(1 << 30)), Lq)
        if rng.random() < 0.5:
            qtr = quantize(qtr, rng, 0.5)
            qtr[qtr < -1000] = -100000.0
            qtr[qtr == 0] = np.float32(-0.0)
        n = int(rng.integers(1, 4))
        tps, ttrs = [], []
        for k in range(n):
            Lt = int(rng.choice(lt_pool))
            if Lq * Lt > 4.2e8:
                Lt = 2101
            tp, ttr = (synth.make_homolog(int(rng.integers(1 << 30)), qp, L=Lt) if rng.random() < 0.6 else
                       synth.make_template(int(rng.integers(1 << 30)), Lt))
            if rng.random() < 0.5:
                ttr = quantize(ttr, rng, 0.5)
                ttr[ttr < -1000] = -100000.0
                ttr[ttr == 0] = np.float32(-0.0)
            if rng.random() < 0.3:
                ttr[rng.integers(0, Lt + 1), rng.integers(0, 7)] = -100000.0
            tps.append(tp)
            ttrs.append(ttr)
        c = capi.Context(local=local, egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"], ss_mode=0)
        c.set_query(qp, qtr)
        ts = c.upload(tps, ttrs)
        so = c.align(ts)
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        want = [orc.align(par, qp, qtr, tps[k], ttrs[k], want_path=True) for k in range(n)]
        for k in range(n):
            a = want[k]
            ok = (a.i2, a.j2) == (int(res["i2"][k]), int(res["j2"][k])) == (int(so["i2"][k]), int(so["j2"][k]))
            ok = ok and np.float32(a.score).tobytes() == np.float32(res["score"][k]).tobytes() == np.float32(so["score"][k]).tobytes()
            ok = ok and int(hits["nsteps"][k]) == a.nsteps and np.float32(hits["score"][k]).tobytes() == np.float32(a.hit_score).tobytes()
            if ok:
                ns, i_s, j_s, st, S = c.hit_path(ts, k)
                ok = (np.array_equal(i_s[1:ns + 1], a.i_steps[1:ns + 1]) and np.array_equal(j_s[1:ns + 1], a.j_steps[1:ns + 1])
                      and np.array_equal(st[1:ns + 1], a.states[1:ns + 1]) and np.array_equal(S[1:ns + 1], a.S[1:ns + 1]))
            cases += 1
            bad += int(not ok)
        # second round on template 0: its first path masked (src/hhviterbirunner.cpp:152-164)
        a = want[0]
        m = orc.exclude_alignment(Lq, tps[0].shape[0] - 1, a.i_steps, a.j_steps, a.nsteps)
        c.set_celloff(ts, 0, m)
        res2 = c.align(ts, celloff=True)
        hits2 = c.hits(ts)
        b = orc.align(par, qp, qtr, tps[0], ttrs[0], celloff=m, want_path=True)
        ok = ((b.i2, b.j2) == (int(res2["i2"][0]), int(res2["j2"][0])) and np.float32(b.score).tobytes() == np.float32(res2["score"][0]).tobytes()
              and int(hits2["nsteps"][0]) == b.nsteps and np.float32(hits2["score"][0]).tobytes() == np.float32(b.hit_score).tobytes())
        cases += 1
        bad += int(not ok)
        ts.free()
        c.close()
    return {"cases": cases, "mismatches": bad}


def fast_family(orc, rng, budget):
    """The opt-in fused-emission build (libhhviterbi_hip_fma.so): bit for bit against the oracle's restatement of its arithmetic
    (emission mode 2), and against the reference's arithmetic (mode 0): how many end points / alignments change, largest score
    difference.  Adversarial inputs as in viterbi_family (ties, dead transitions, tiny templates), no masks."""
    t_end, cases, bad, idx_changed, worst, worst_hit = time.time() + budget, 0, 0, 0, 0.0, 0.0
    while time.time() < t_end:
        Lq = int(rng.choice([5, 63, 64, 65, 100, 300, 320, 321, 400]))
        local = int(rng.integers(0, 2))
        par = po.make_params(local=local, egq=float(rng.choice([0.0, 0.2])), egt=float(rng.choice([0.0, 0.1])),
                             shift=float(rng.choice([-0.03, 0.0])), ss_mode=0)
        qp, qtr = synth.make_query(int(rng.integers(1 << 30)), Lq)
        n = int(rng.integers(4, 24))
        tps, ttrs = [], []
        for k in range(n):
            Lt = int(rng.choice([1, 3, 31, 64, 130, 257, 300]))
            tp, ttr = (synth.make_homolog(int(rng.integers(1 << 30)), qp, L=Lt) if rng.random() < 0.6 and Lq > 4 else
                       synth.make_template(int(rng.integers(1 << 30)), Lt))
            if rng.random() < 0.3:
                ttr = quantize(ttr, rng, 0.5)
                ttr[ttr < -1000] = -100000.0
            tps.append(tp)
            ttrs.append(ttr)
        c = capi.Context(local=local, egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"], ss_mode=0,
                         lib_path=capi.FMA_LIB_PATH)
        c.set_query(qp, qtr)
        ts = c.upload(tps, ttrs)
        res = c.align(ts, backtrace=True)
        hits = c.hits(ts)
        try:
            orc.set_emission_mode(2)
            own = [orc.align(par, qp, qtr, tps[k], ttrs[k], want_path=True) for k in range(n)]
        finally:
            orc.set_emission_mode(0)
        ref = [orc.align(par, qp, qtr, tps[k], ttrs[k], want_path=True) for k in range(n)]
        for k in range(n):
            a, r = own[k], ref[k]
            ok = (a.i2, a.j2) == (int(res["i2"][k]), int(res["j2"][k])) and np.float32(a.score).tobytes() == np.float32(res["score"][k]).tobytes()
            ok = ok and int(hits["nsteps"][k]) == a.nsteps and np.float32(hits["score"][k]).tobytes() == np.float32(a.hit_score).tobytes()
            cases += 1
            bad += int(not ok)
            same = (a.i2, a.j2, a.nsteps) == (r.i2, r.j2, r.nsteps) and np.array_equal(a.i_steps[1:a.nsteps + 1], r.i_steps[1:r.nsteps + 1]) \
                and np.array_equal(a.states[1:a.nsteps + 1], r.states[1:r.nsteps + 1])
            idx_changed += int(not same)
            if np.isfinite(a.score) and np.isfinite(r.score):
                worst = max(worst, abs(float(a.score) - float(r.score)))
            if same:
                worst_hit = max(worst_hit, abs(float(a.hit_score) - float(r.hit_score)))
        ts.free()
        c.close()
    return {"cases": cases, "mismatches_vs_own_oracle": bad, "alignments_changed_vs_reference_arithmetic": idx_changed,
            "max_abs_viterbi_score_diff": worst, "max_abs_hit_score_diff_same_alignment": worst_hit}


def prefilter_family(orc, rng, budget):
    u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_ubyte))
    t_end, cases, bad = time.time() + budget, 0, 0
    diag = []
    c = capi.Context()
    while time.time() < t_end:
        Lq = int(rng.choice([1, 31, 32, 33, 64, 65, 255, 300, 512, 513, 641, 900]))
        off = int(rng.choice([50, 0, 128, 20]))
        mode = rng.integers(0, 3)
        prof = (rng.integers(0, 256, (220, Lq)) if mode == 0 else np.clip(rng.normal(off - 4, 14, (220, Lq)), 0, 255)).astype(np.uint8)
        if mode == 2:
            prof[rng.integers(0, 220, Lq), np.arange(Lq)] = min(255, off + 127)   # the largest byte the fast path takes
        n = int(rng.integers(1, 40))
        lens = rng.integers(0, 400, n)
        lens[rng.integers(0, n)] = 1
        offs = np.zeros(n + 1, np.int64)
        offs[1:] = np.cumsum(lens)
        seqs = rng.integers(0, 220, offs[-1]).astype(np.uint8)
        # every second sequence a homolog: it follows the profile's best states along a diagonal, with insertions, deletions and
        # repeats - high scores up to the cap, where the Smith-Waterman kernel's lazy-F correction (a prefix scan for gap open >=
        # gap extend, the reference's loop otherwise) has 40-60 rows to repair behind every strong cell
        best = prof[:219].argmax(axis=0).astype(np.uint8)
        for k in range(0, n, 2):
            L, q, t = int(lens[k]), int(rng.integers(0, max(1, Lq // 2))), 0
            frac = float(rng.choice([0.3, 0.7, 0.95]))
            while t < L:
                r = rng.random()
                if r < 0.04:
                    q += int(rng.integers(1, 15))
                elif r < 0.08:
                    t += int(rng.integers(1, 6))
                    continue
                if q >= Lq:
                    q = int(rng.integers(0, max(1, Lq // 2)))
                if rng.random() < frac:
                    seqs[offs[k] + t] = best[q]
                t += 1
                q += 1
        go, ge = int(rng.choice([24, 20, 5, 0, 40, 9])), int(rng.choice([4, 1, 0, 9]))
        db = c.prefilter_upload_db(seqs, offs)
        ung = c.prefilter_scores(db, prof, off, gapped=False)
        gap = c.prefilter_scores(db, prof, off, gapped=True, gap_init=go, gap_extend=ge)
        for k in range(n):
            s = np.ascontiguousarray(seqs[offs[k]:offs[k + 1]]) if lens[k] else np.zeros(1, np.uint8)
            w_u = orc.lib.hho_ungapped_score(u8(prof), Lq, u8(s), int(lens[k]), off)
            w_g = orc.lib.hho_sw_score(u8(prof), Lq, u8(s), int(lens[k]), go, ge, off, 32)
            cases += 1
            if w_u != ung[k] or w_g != gap[k]:
                bad += 1
                if len(diag) < 8:
                    diag.append({"Lq": Lq, "off": off, "mode": int(mode), "len": int(lens[k]), "go": go, "ge": ge, "k": k, "n": n,
                                 "ung": [int(w_u), int(ung[k])], "gap": [int(w_g), int(gap[k])], "pmax": int(prof.max())})
        c.prefilter_free_db(db)
    c.close()
    return {"cases": cases, "mismatches": bad, "diag": diag}


def mac_family(orc, rng, budget):
    import test_mac as T
    t_end, cases, bad = time.time() + budget, 0, 0
    diag = []
    c = capi.Context()
    while time.time() < t_end:
        Lq = int(rng.choice([2, 3, 40, 64, 65, 129, 200]))
        local = int(rng.integers(0, 2))
        mact = float(rng.choice([0.3501, 0.0, 0.9, 0.05]))
        qp, qtr = synth.make_query(int(rng.integers(1 << 30)), Lq)
        q_lin = T.lin_query(qtr)
        tps, tls, masks, want = [], [], [], []
        lists = bool(rng.integers(0, 2))   # every other batch with the -o_matrices lists (hhv_mac_set_lists)
        for k in range(int(rng.integers(1, 6))):
            Lt = int(rng.choice([1, 2, 63, 64, 65, 128, 190]))
            if rng.random() < 0.15:     # the other two length classes of the launch (LDS row state only; rows in global memory)
                Lt = int(rng.choice([790, 805, 1000, 2046, 2047, 2300]))
            tp, ttr = (synth.make_homolog(int(rng.integers(1 << 30)), qp, L=Lt) if rng.random() < 0.6 and Lq > 4 else
                       synth.make_template(int(rng.integers(1 << 30)), Lt))
            t_lin = T.lin_template(ttr)
            dens = rng.choice([0.0, 0.1, 0.6, 0.97])
            m = (rng.random((Lq + 1, Lt + 1)) < dens).astype(np.uint8)
            m[0, :] = 0
            m[:, 0] = 0
            o = po._mac_buffers(Lq, Lt)
            L = orc.lib
            V = C.c_void_p
            L.hho_mac_forward.argtypes = [V, V, C.c_int, V, V, C.c_int, C.c_int, C.c_float, V, V, V, V]
            L.hho_mac_backward.argtypes = [V, V, C.c_int, V, V, C.c_int, C.c_int, C.c_float, V, V, C.c_double, V]
            L.hho_mac_dp.argtypes = [V, V, C.c_int, C.c_int, C.c_int, C.c_float, V, V, V]
            L.hho_mac_backtrace.argtypes = [V, V, V, V, C.c_int, C.c_int, C.c_int, C.c_int] + [V] * 8
            L.hho_mac_forward(qp.ctypes.data, q_lin.ctypes.data, Lq, tp.ctypes.data, t_lin.ctypes.data, Lt, local, -0.03,
                              m.ctypes.data, o.forward.ctypes.data, o.scale.ctypes.data, C.addressof(o.Pforward))
            o.posterior[:] = o.forward
            o.fwd_list = np.zeros((Lq + 1, Lt + 1), np.float32)
            o.bwd_list = np.zeros((Lq + 1, Lt + 1), np.float32)
            L.hho_mac_forward_list.argtypes = [V, C.c_int, C.c_int, V, C.c_double, V]
            L.hho_mac_backward_list.argtypes = [V, V, C.c_int, V, V, C.c_int, C.c_int, C.c_float, V, V, C.c_double, V, V]
            L.hho_mac_forward_list(o.forward.ctypes.data, Lq, Lt, o.scale.ctypes.data, o.Pforward.value, o.fwd_list.ctypes.data)
            L.hho_mac_backward_list(qp.ctypes.data, q_lin.ctypes.data, Lq, tp.ctypes.data, t_lin.ctypes.data, Lt, local, -0.03,
                                    m.ctypes.data, o.scale.ctypes.data, o.Pforward.value, o.posterior.ctypes.data, o.bwd_list.ctypes.data)
            i2, j2, ns, mc = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            L.hho_mac_dp(o.posterior.ctypes.data, m.ctypes.data, Lq, Lt, local, mact, o.bmm.ctypes.data, C.addressof(i2), C.addressof(j2))
            L.hho_mac_backtrace(o.bmm.ctypes.data, o.posterior.ctypes.data, qp.ctypes.data, tp.ctypes.data, Lq, Lt, i2.value,
                                j2.value, o.i_steps.ctypes.data, o.j_steps.ctypes.data, o.states.ctypes.data, o.S.ctypes.data,
                                o.P.ctypes.data, C.addressof(ns), C.addressof(mc), C.addressof(o.sum_of_probs))
            o.ns, o.i2v, o.j2v, o.Pf = ns.value, i2.value, j2.value, o.Pforward.value
            tps.append(tp)
            tls.append(t_lin)
            masks.append(m)
            want.append(o)
        c.mac_set_lists(lists)
        ms = c.mac_realign(qp, q_lin, tps, tls, masks, local=local, mact=mact)
        for k, o in enumerate(want):
            h = ms.hits[k]
            ok = np.float64(h["Pforward"]).tobytes() == np.float64(o.Pf).tobytes() or (np.isnan(o.Pf) and np.isnan(h["Pforward"]))
            post = ms.posterior(k)
            # a mask that leaves no path gives Pforward = 0 and NaN posteriors in the reference too: NaN == NaN here
            ok = ok and np.array_equal(post[1:, 1:], o.posterior[1:, 1:], equal_nan=True)
            ok = ok and (int(h["nsteps"]), int(h["i2"]), int(h["j2"])) == (o.ns, o.i2v, o.j2v)
            if ok and o.ns:
                i_s, j_s, st, S, P = ms.path(k)
                ok = np.array_equal(i_s[1:], o.i_steps[1:o.ns + 1]) and np.array_equal(P[1:], o.P[1:o.ns + 1], equal_nan=True)
            lists_ok = True
            if lists:
                wl = [po.mac_plane_to_list(o.fwd_list), po.mac_plane_to_list(o.bwd_list),
                      po.mac_posterior_list(o.posterior, masks[k], o.i_steps, o.j_steps, o.ns)]
                for w in range(3):
                    g = ms.list(k, w)
                    lists_ok = lists_ok and np.array_equal(g[0], wl[w][0]) and np.array_equal(g[1], wl[w][1]) and g[2].tobytes() == wl[w][2].tobytes()
                ok = ok and lists_ok
            cases += 1
            if not ok:
                bad += 1
                if len(diag) < 8:
                    diag.append({"Lq": Lq, "Lt": int(tps[k].shape[0] - 1), "local": local, "lists_ok": lists_ok, "mact": mact, "dens": float(masks[k].mean()),
                                 "Pf": [o.Pf, float(h["Pforward"])], "post_eq": bool(np.array_equal(post[1:, 1:], o.posterior[1:, 1:], equal_nan=True)),
                                 "ends": [[o.ns, o.i2v, o.j2v], [int(h["nsteps"]), int(h["i2"]), int(h["j2"])]],
                                 "npost_diff": int((post[1:, 1:] != o.posterior[1:, 1:]).sum())})
        ms.free()
    c.close()
    return {"cases": cases, "mismatches": bad, "diag": diag}


def prepare_family(orc, rng, budget):
    z = np.load(os.path.join(ROOT, "tests", "golden", "gonnet_pb_R.npz"))
    pb, R = z["pb"], z["R"]
    t_end, cases, bad = time.time() + budget, 0, 0
    c = capi.Context(local=1)
    qp, qtr = synth.make_query(3, 50)
    c.set_query(qp, qtr)
    while time.time() < t_end:
        n = int(rng.integers(1, 10))
        raws = []
        for k in range(n):
            L = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 447, 448, 500, 1301]))
            f, tr, neff, nh = synth.make_raw_hmm(int(rng.integers(1 << 30)), L)
            if rng.random() < 0.3:      # columns with a single residue / all-star transitions
                j = int(rng.integers(1, L + 1))
                f[j] = 2.0 ** -99.999
                f[j, rng.integers(0, 20)] = 1.0
                tr[j - 1] = [0.0, -99.999, -99.999, 0.0, -99.999, 0.0, -99.999]
            if rng.random() < 0.2:
                neff[:, 0] = 1.0        # Neff_M = 1: (nM - 1) = 0 in the transition pseudocounts
            raws.append((f, tr, neff, nh))
        pcm = int(rng.integers(0, 4))   # (pcm 3 takes its admixture constant from pcb: 0.793 + 0.048 (pcb - 10) is in [0, 1] for these)
        pc = np.array([pcm, float(rng.choice([1.0, 0.4, 0.0, 0.85, 1.7 if pcm == 2 else 1.0])), float(rng.choice([1.5, 0.5, 4.0])),
                       float(rng.choice([1.0, 1.0, 0.7, 1.6]))], np.float32)
        gap = np.array([rng.choice([0.15, 1.0]), rng.choice([1.0, 0.3, 2.0]), 0.6, rng.choice([0.6, 1.0]), 0.6, 0.6,
                        rng.choice([1.0, 0.0, 2.5])], np.float32)
        cs = int(rng.integers(0, 4))
        q_pav = rng.dirichlet(np.ones(20) * 5).astype(np.float32)
        raw, Ls = c.upload_raw([r[0] for r in raws], [r[1] for r in raws], [r[2] for r in raws], [r[3] for r in raws])
        ts = c.prepare(raw, Ls, capi.prep_params(pb, R, gap=tuple(gap), pc=tuple(pc), columnscore=cs), q_pav)
        for k, (f, tr, neff, nh) in enumerate(raws):
            p, tro, pv = po.oracle_prepare(orc, 1, f, tr, neff, nh, pb, R, q_pav=q_pav, gap=gap, pc=pc, columnscore=cs)
            want = capi.pack_profile(np.ascontiguousarray(p[:-1]), tro, index=k)
            cases += 1
            bad += int(not np.array_equal(c.records_of(ts, k).view(np.uint32), want.view(np.uint32)))
        c.rawset_free(raw)
        ts.free()
    c.close()
    return {"cases": cases, "mismatches": bad}


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    orc = po.Oracle()
    fam = sys.argv[3] if len(sys.argv) > 3 else "all"
    if fam != "all":
        print(json.dumps({fam: {"prefilter": prefilter_family, "mac": mac_family, "viterbi": viterbi_family, "prepare": prepare_family,
                                "fast": fast_family, "long": long_family, "ss": lambda o, r, b: viterbi_family(o, r, b, with_ss=True)}[fam](orc, rng, budget)}))
        return
    out = {"seconds_per_family": budget,
           "viterbi_backtrace_celloff": viterbi_family(orc, rng, budget),
           "viterbi_secondary_structure": viterbi_family(orc, rng, budget, with_ss=True),
           "viterbi_long_profiles": long_family(orc, rng, budget),
           "prefilter": prefilter_family(orc, rng, budget),
           "mac_realign": mac_family(orc, rng, budget),
           "prepare": prepare_family(orc, rng, budget)}
    if os.path.exists(capi.FMA_LIB_PATH):
        out["fast_mode_opt_in_build"] = fast_family(orc, rng, budget)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
