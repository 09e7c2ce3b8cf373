#!/usr/bin/env python3
"""tools/bench_dropin.py <n_templates> <threads> [L] -- wall time of ViterbiRunner::alignment end to end (template
text parsing + PrepareTemplateHMM on the host included, exactly what a caller of the reference pays): the reference's own
src/hhviterbirunner.cpp against the drop-in translation unit hh-suite_amd/dropin/hhviterbirunner_hip.cpp, both through
oracle/ref_runner_harness.cpp on the same .hhm texts.  Run on the GPU box (needs oracle/_ref/libhhref_dropin.so)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import hhm_text
from test_dropin_runner import run, compare, cache_clear


def canonical(r):
    """hits in (irep, entry) order: with several threads the reference returns them in thread-major order"""
    order = sorted(range(len(r[0])), key=lambda k: (r[0][k].irep, r[0][k].entry))
    return ([r[0][k] for k in order],) + tuple(a[order] for a in r[1:])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    L = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    uniq = min(n, 400)
    qf = hhm_text.random_columns(1, L)
    q = hhm_text.hhm_text("query", qf, 1)
    base = []
    for k in range(uniq):
        f = hhm_text.mutate_columns(k, qf, 0.4) if k % 50 == 0 else hhm_text.random_columns(100 + k, L)
        base.append(f)
    texts, names = [], []
    for k in range(n):
        names.append("b%06d" % k)
        texts.append(hhm_text.hhm_text(names[-1], base[k % uniq], k % uniq))
    out = {"n_templates": n, "L": L, "threads": threads}
    for altali in (1, 4):
        kw = dict(altali=altali, maxres=L + 100, path_cap=2 * L + 10)
        res = {}
        cache_clear()
        for which, tag in (("hip", "dropin_cold"), ("cpu", "cpu"), ("hip", "dropin_warm"), ("cpu", "cpu"), ("hip", "dropin_warm")):
            r = run(which, q, texts, names, threads=threads, **kw)
            res[tag] = (run.last_alignment_seconds, r)     # the ViterbiRunner::alignment call alone
        compare(canonical(res["cpu"][1]), canonical(res["dropin_cold"][1]))
        compare(canonical(res["cpu"][1]), canonical(res["dropin_warm"][1]))
        out["altali%d" % altali] = {"reference_s": round(res["cpu"][0], 4), "dropin_cold_cache_s": round(res["dropin_cold"][0], 4),
                                    "dropin_warm_cache_s": round(res["dropin_warm"][0], 4),
                                    "hits": len(res["cpu"][1][0]), "hits_identical": True, "cells": int(n) * L * L}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
