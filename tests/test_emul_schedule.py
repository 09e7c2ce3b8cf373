"""Systolic schedule on the CPU: the product's per-lane code (hh-suite_amd/csrc/viterbi_lane.h)
stepped in lock-step for 64 lanes by tests/emul/wave_emul.cpp must reproduce the oracle bit for bit
(scores up to the sign of zero, endpoints and every backtrace byte exactly)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from common import same_float, workload
from pyhhv import pack
from pyoracle import make_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "emul", "libwave_emul.so")


class TR(C.Structure):
    _fields_ = [("score", C.c_float), ("i2", C.c_int), ("j2", C.c_int), ("tid", C.c_int)]


@pytest.fixture(scope="module")
def emul():
    if not os.path.exists(SO):
        subprocess.check_call(["make", "-C", ROOT, "emul"])
    lib = C.CDLL(SO)
    lib.hhv_emul_wave.argtypes = [C.c_int] * 4 + [C.c_void_p, C.c_void_p, C.c_long, C.c_float, C.c_float, C.c_float,
                                                  C.c_int, C.POINTER(TR), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    return lib


def ss_operands(par, ss, Lq, R, P):
    """Premultiplied table + per-row offsets + (shift, mask) exactly like hhv_api.cpp::ensure_ss."""
    T = {4: ss.S33, 2: ss.S73, 1: ss.S37}[ss.mode].reshape(-1)
    tab = (np.float32(par["ssw"]) * T).astype(np.float32)
    off = np.zeros(P * 64 * R, dtype=np.int32)
    for i in range(1, Lq + 1):
        pr, cf, ds = int(ss.q_pred[i]), int(ss.q_conf[i]), int(ss.q_dssp[i])
        off[i - 1] = {4: (pr * 11 + cf) * 44, 2: ds * 44, 1: (pr * 11 + cf) * 8}[ss.mode]
    shift, mask = (22, 7) if ss.mode == 1 else (16, 0x3F)
    return tab, off, shift, mask


def run(emul, par, qf, qtr, tps, ttrs, want_bt, bt_in=None, ss=None, t_ss=None):
    Lq = qf.shape[0] - 1
    R, P = pack.strips_for(Lq)
    qpack = pack.pack_query(qf, qtr, R, P)
    rec, off = pack.pack_stream(tps, ttrs, t_ss)
    M = rec.shape[0]
    n = len(tps)
    res = (TR * n)()
    bt = np.zeros((P, M, 64), dtype=np.uint64) if bt_in is None else bt_in
    ssargs = (None, None, 0, 0)
    if ss is not None:
        tab, qoff, shift, mask = ss_operands(par, ss, Lq, R, P)   # keep the arrays alive across the call
        ssargs = (tab.ctypes.data, qoff.ctypes.data, shift, mask)
    em = emul.hhv_emul_wave(R, par["local"], int(want_bt), int(bt_in is not None), qpack.ctypes.data, rec.ctypes.data,
                            M, par["egq"], par["egt"], par["shift"], Lq, res, n, bt.ctypes.data, P, *ssargs)
    assert em == n
    run.mm_mode = emul.hhv_emul_bt_mm_mode(R, par["local"], int(bt_in is not None), int(ss is not None))  # how to decode bt
    return res, bt, off, R


@pytest.mark.parametrize("case", range(20))
def test_schedule_matches_oracle(emul, oracle, case):
    rng = np.random.default_rng(case)
    Lq = int(rng.integers(5, 330)) if case < 14 else [431, 512, 321, 700, 1000, 1281][case - 14]
    par = make_params(local=case % 2, egq=0.0 if case % 4 < 2 else 0.3, egt=0.0 if case % 4 < 2 else 0.1)
    n = int(rng.integers(1, 7))
    qf, qtr, tps, ttrs = workload(case, Lq, n, 1, 200)
    for want_bt in (0, 1):
        res, bt, off, R = run(emul, par, qf, qtr, tps, ttrs, want_bt)
        for e in range(n):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=True)
            assert same_float(a.score, res[e].score) and (a.i2, a.j2) == (res[e].i2, res[e].j2), (case, e)
            if want_bt:
                m = pack.bt_to_matrix(bt.view(np.uint8), int(off[e]), Lq, tps[e].shape[0] - 1, R, mm_mode=run.mm_mode)
                assert np.array_equal(m[1:, 1:], a.bt[1:, 1:])


def test_schedule_celloff(emul, oracle):
    for local in (0, 1):
        par = make_params(local=local)
        qf, qtr, tps, ttrs = workload(40 + local, 150, 3, 80, 170, homolog_every=1)
        Lq = 150 if local else 400
        qf, qtr, tps, ttrs = workload(40 + local, Lq, 3, 80, 170, homolog_every=1)
        R, P = pack.strips_for(Lq)
        rec, off = pack.pack_stream(tps, ttrs)
        bt = np.zeros((P, rec.shape[0], 64), dtype=np.uint64)
        masks = []
        for e in range(3):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_path=True)
            m = oracle.exclude_alignment(Lq, tps[e].shape[0] - 1, a.i_steps, a.j_steps, a.nsteps)
            masks.append(m)
            pack.matrix_to_bt(m, int(off[e]), R, bt.view(np.uint8))
        res, bt2, off, R = run(emul, par, qf, qtr, tps, ttrs, 1, bt_in=bt)
        for e in range(3):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=masks[e], want_bt=True)
            assert same_float(a.score, res[e].score) and (a.i2, a.j2) == (res[e].i2, res[e].j2)
            m = pack.bt_to_matrix(bt2.view(np.uint8), int(off[e]), Lq, tps[e].shape[0] - 1, R, mm_mode=run.mm_mode)
            assert np.array_equal(m[1:, 1:], a.bt[1:, 1:])


def test_schedule_secondary_structure(emul, oracle):
    """...AndSS variants for the three ss_hmm_modes, single and multi pass."""
    from pyoracle import SSInfo
    rng = np.random.default_rng(5)
    S73 = rng.normal(0, 1, (8, 4, 11)).astype(np.float32)
    S33 = rng.normal(0, 1, (4, 11, 4, 11)).astype(np.float32)
    S37 = rng.normal(0, 1, (4, 11, 8)).astype(np.float32)
    for Lq, local in ((80, 1), (330, 0)):
        par = make_params(local=local, ss_mode=2)
        qf, qtr, tps, ttrs = workload(9 + Lq, Lq, 3, 40, 100, homolog_every=1)
        for mode in (1, 2, 4):
            ss = SSInfo(mode, rng.integers(0, 4, Lq + 1), rng.integers(0, 11, Lq + 1), rng.integers(0, 8, Lq + 1),
                        S73, S33, S37)
            t_ss = [(rng.integers(0, 4, p.shape[0]), rng.integers(0, 11, p.shape[0]), rng.integers(0, 8, p.shape[0]))
                    for p in tps]
            res, bt, off, R = run(emul, par, qf, qtr, tps, ttrs, 1, ss=ss, t_ss=t_ss)
            for e in range(3):
                a = oracle.align(par, qf, qtr, tps[e], ttrs[e], ss=ss, t_ss=t_ss[e], want_bt=True)
                assert same_float(a.score, res[e].score) and (a.i2, a.j2) == (res[e].i2, res[e].j2), (Lq, mode, e)
                m = pack.bt_to_matrix(bt.view(np.uint8), int(off[e]), Lq, tps[e].shape[0] - 1, R, mm_mode=run.mm_mode)
                assert np.array_equal(m[1:, 1:], a.bt[1:, 1:])


@pytest.mark.parametrize("local", [0, 1])
def test_sign_bit_flags_with_exact_ties_and_negative_zero(emul, oracle, local):
    """The 64-lane backtrace variants take their compare bits from the SIGN of a difference (viterbi_lane.h BT_PAIR_SIGN):
    exact as long as no DP state is -0.  Transition scores on a 0.5 grid (exact ties between candidates, sums that cancel
    to +0), transitions that ARE -0.0f, zero end-gap penalties (the reference's boundary -j * egt is -0 there) and profile
    columns repeated so that whole cells tie: every backtrace byte must still be the reference's."""
    from pyhhv import synth
    par = make_params(local=local, egq=0.0, egt=0.0)
    for Lq in (37, 300):
        qf, qtr = synth.make_query(7700 + Lq, Lq)
        qtr = (np.round(qtr / 0.5) * 0.5).astype(np.float32)
        qtr[qtr == 0] = np.float32(-0.0)
        qf[5:9] = qf[4]
        tps, ttrs = [], []
        for e in range(4):
            Lt = [33, 120, 64, 7][e]
            p, tr = synth.make_homolog(7800 + e, qf, L=Lt) if e % 2 == 0 else synth.make_template(7800 + e, Lt)
            tr = (np.round(tr / 0.5) * 0.5).astype(np.float32)
            tr[tr == 0] = np.float32(-0.0)
            if Lt > 12:
                p[8:12] = p[7]
            tps.append(p)
            ttrs.append(tr)
        res, bt, off, R = run(emul, par, qf, qtr, tps, ttrs, 1)
        assert run.mm_mode == pack.BT_MM_FIRST_EQUAL_NEG
        for e in range(4):
            a = oracle.align(par, qf, qtr, tps[e], ttrs[e], want_bt=True)
            assert same_float(a.score, res[e].score) and (a.i2, a.j2) == (res[e].i2, res[e].j2), (Lq, e)
            m = pack.bt_to_matrix(bt.view(np.uint8), int(off[e]), Lq, tps[e].shape[0] - 1, R, mm_mode=run.mm_mode)
            assert np.array_equal(m[1:, 1:], a.bt[1:, 1:]), (Lq, e)
