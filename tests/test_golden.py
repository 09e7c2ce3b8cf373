"""Committed golden vectors (tests/golden/viterbi_golden.json, produced by the reference itself via
tests/golden/make_golden.py): the oracle must reproduce them on any box (CPU test), and the HIP
engine must reproduce them through the C ABI (GPU test).  /root/reference is not needed at run time."""
import hashlib
import json
import os

import numpy as np
import pytest

from golden_cases import build_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "viterbi_golden.json")


def load():
    with open(GOLD) as f:
        return json.load(f)["cases"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def f32_bits(x):
    return int(np.float32(x).view(np.uint32))


def bits_equal_mod_zero_sign(a_bits, b_bits):
    return a_bits == b_bits or ((a_bits | b_bits) & 0x7FFFFFFF) == 0


@pytest.mark.parametrize("idx", range(9))
def test_oracle_reproduces_golden(oracle, idx):
    case = load()[idx]
    par, qf, qtr, tps, ttrs, masks = build_case(case["spec"])
    for e, g in enumerate(case["templates"]):
        assert tps[e].shape[0] - 1 == g["Lt"]
        a = oracle.align(par, qf, qtr, tps[e], ttrs[e], celloff=None if masks is None else masks[e], want_path=True)
        assert f32_bits(a.score) == g["score_bits"] and (a.i2, a.j2) == (g["i2"], g["j2"])
        assert sha(a.bt[1:, 1:]) == g["bt_sha256"]
        ns = g["nsteps"]
        assert (a.nsteps, a.matched_cols) == (ns, g["matched_cols"])
        assert list(a.i_steps[1:ns + 1]) == g["i_steps"] and list(a.j_steps[1:ns + 1]) == g["j_steps"]
        assert list(a.states[1:ns + 1]) == g["states"]
        assert sha(a.S[1:ns + 1]) == g["S_sha256"] and f32_bits(a.hit_score) == g["hit_score_bits"]


@pytest.mark.gpu
@pytest.mark.parametrize("idx", range(9))
def test_gpu_reproduces_golden(idx):
    from pyhhv import capi
    case = load()[idx]
    par, qf, qtr, tps, ttrs, masks = build_case(case["spec"])
    c = capi.Context(local=par["local"], egq=par["egq"], egt=par["egt"], shift=par["shift"], corr=par["corr"],
                     ssw=par["ssw"], ss_mode=par["ss_mode"])
    c.set_query(qf, qtr)
    ts = c.upload(tps, ttrs)
    if masks is not None:
        for e, m in enumerate(masks):
            c.set_celloff(ts, e, m)
    res = c.align(ts, backtrace=True, celloff=masks is not None)
    hits = c.hits(ts)
    for e, g in enumerate(case["templates"]):
        assert bits_equal_mod_zero_sign(f32_bits(res["score"][e]), g["score_bits"])
        assert (int(res["i2"][e]), int(res["j2"][e])) == (g["i2"], g["j2"])
        assert sha(c.backtrace_matrix(ts, e)[1:, 1:]) == g["bt_sha256"]
        ns, i_s, j_s, st, S = c.hit_path(ts, e)
        assert (ns, int(hits["matched_cols"][e])) == (g["nsteps"], g["matched_cols"])
        assert list(i_s[1:ns + 1]) == g["i_steps"] and list(j_s[1:ns + 1]) == g["j_steps"]
        assert list(st[1:ns + 1]) == g["states"]
        assert sha(S[1:ns + 1]) == g["S_sha256"]
        assert bits_equal_mod_zero_sign(f32_bits(hits["score"][e]), g["hit_score_bits"])
    ts.free()
    c.close()
