#!/usr/bin/env python3
"""tools/bench_dropin.py <n_templates> <threads> [L] -- wall time of ViterbiRunner::alignment end to end (template
text parsing + PrepareTemplateHMM on the host included, exactly what a caller of the reference pays): the reference's own
src/hhviterbirunner.cpp against the drop-in translation unit hh-suite_amd/dropin/hhviterbirunner_hip.cpp, both through
oracle/ref_runner_harness.cpp on the same .hhm texts, hits compared.  Run on the GPU box (needs
oracle/_ref/libhhref_dropin.so)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def canonical(r):
    """hits in (irep, entry) order: with several threads the reference returns them in thread-major order"""
    order = sorted(range(len(r[0])), key=lambda k: (r[0][k].irep, r[0][k].entry))
    return ([r[0][k] for k in order],) + tuple(a[order] for a in r[1:])


class _Stderr:
    """the C-level stderr of this process into a file for the length of a `with` block (HHV_DROPIN_TIMING's report)"""
    def __enter__(self):
        import tempfile
        sys.stderr.flush()
        self.f = tempfile.TemporaryFile()
        self.saved = os.dup(2)
        os.dup2(self.f.fileno(), 2)
        return self

    def __exit__(self, *a):
        os.dup2(self.saved, 2)
        os.close(self.saved)
        self.f.seek(0)
        self.text = self.f.read().decode(errors="replace")
        self.f.close()


def _phases(text):
    """the last 'in ms:' line of the drop-in's phase timer -> {label: ms}"""
    lines = [l for l in text.splitlines() if "hhviterbirunner_hip:   in ms:" in l]
    if not lines:
        return None
    out = {}
    for item in lines[-1].split("in ms:")[1].split(","):
        item = item.strip()
        if item:
            label, ms = item.rsplit(" ", 1)
            out[label] = float(ms)
    return out


def run(n=4000, threads=32, L=300, altalis=(1, 4), phases=False):
    import hhm_text
    from test_dropin_runner import run as runner, compare, cache_clear
    uniq = min(n, 400)
    qf = hhm_text.random_columns(1, L)
    q = hhm_text.hhm_text("query", qf, 1)
    base = []
    for k in range(uniq):
        f = hhm_text.mutate_columns(k, qf, 0.4) if k % 50 == 0 else hhm_text.random_columns(100 + k, L)
        base.append(hhm_text.hhm_text("@NAME@", f, k))
    names = ["b%06d" % k for k in range(n)]
    texts = [base[k % uniq].replace(b"@NAME@", names[k].encode()) for k in range(n)]
    out = {"n_templates": n, "L": L, "threads": threads}
    for altali in altalis:
        kw = dict(altali=altali, maxres=L + 100, path_cap=2 * L + 10)
        res = {}
        cache_clear()
        for which, tag in (("hip", "dropin_cold"), ("cpu", "cpu"), ("hip", "dropin_warm"), ("cpu", "cpu"), ("hip", "dropin_warm")):
            if phases and which == "hip":
                os.environ["HHV_DROPIN_TIMING"] = "1"
                with _Stderr() as cap:
                    r = runner(which, q, texts, names, threads=threads, **kw)
                del os.environ["HHV_DROPIN_TIMING"]
                res[tag + "_phases_ms"] = _phases(cap.text)
            else:
                r = runner(which, q, texts, names, threads=threads, **kw)
            if tag == "dropin_warm":
                res.setdefault("warm_all_s", []).append(round(runner.last_alignment_seconds, 4))
                if "dropin_warm" in res and res["dropin_warm"][0] <= runner.last_alignment_seconds:
                    continue      # (the faster of the warm calls stands: the host side of a call varies by several ms - fresh pages for
                                  #  20 000 Hit objects after the reference's run in between, the OpenMP team's wake-up)
                if phases:
                    res["dropin_warm_best_phases_ms"] = res.get("dropin_warm_phases_ms")
            res[tag] = (runner.last_alignment_seconds, r)     # the ViterbiRunner::alignment call alone
        compare(canonical(res["cpu"][1]), canonical(res["dropin_cold"][1]))
        compare(canonical(res["cpu"][1]), canonical(res["dropin_warm"][1]))
        out["altali%d" % altali] = {"reference_s": round(res["cpu"][0], 4), "dropin_cold_cache_s": round(res["dropin_cold"][0], 4),
                                    "dropin_warm_cache_s": round(res["dropin_warm"][0], 4),
                                    "hits": len(res["cpu"][1][0]), "hits_identical": True, "cells": int(n) * L * L}
        if phases:
            out["altali%d" % altali]["cold_phases_ms"] = res.get("dropin_cold_phases_ms")
            out["altali%d" % altali]["warm_phases_ms"] = res.get("dropin_warm_best_phases_ms") or res.get("dropin_warm_phases_ms")
        out["altali%d" % altali]["dropin_warm_calls_s"] = res.get("warm_all_s")
    cache_clear()
    return out


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    print(json.dumps(run(*a)))
