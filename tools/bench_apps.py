#!/usr/bin/env python3
"""tools/bench_apps.py [n_templates] [threads] [n_queries] -- wall time of the reference's own command-line programs, built by
oracle/Makefile from the reference's sources (`*_cpu`) and with the three translation units of hh-suite_amd/dropin/ swapped in
(`*_hip`), on one synthetic ffindex database (templates and queries of 300 columns):
  hhsearch      -i query.hhm   -d db -cpu <threads>          one query, the threads work on its templates
  hhsearch_omp  -i queries     -d db -cpu <min(threads, nq)> n_queries queries, one thread each (src/hhblits_omp.cpp)
Whole-process wall time (start-up, database open, query read, search, realignment, output files); outputs compared.
Run on the GPU box."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import hhm_text
from test_dropin_apps import BIN, build_db, compare_outputs, read_ffindex, run_app, write_ffindex


def timed(fn, *a):
    t0 = time.time()
    r = fn(*a)
    return time.time() - t0, r


def run_omp(binary, args, prefix):
    cmd = [os.path.join(BIN, binary)] + args + ["-o", prefix + "_hhr", "-v", "1"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=3000)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    return read_ffindex(prefix + "_hhr")


def sidecar_processes(n=4000, threads=16, L=300):
    """One cold PROCESS each of the reference's hhsearch and of the same program with the replaced translation units: without
    the binary sidecar of the hhm database (every template parsed from its text by HMM::Read, like the reference), the run that
    writes the sidecar, and a cold process that finds it (what N1 is for).  Whole-process wall time and the drop-in's own
    account of ViterbiRunner::alignment (HHV_DROPIN_TIMING)."""
    import re
    qf = hhm_text.random_columns(900, L)
    query = hhm_text.hhm_text("query00", qf, 900)
    uniq = min(n, 400)
    base_txt = []
    for k in range(uniq):
        f = hhm_text.mutate_columns(k, qf, 0.4) if k % 20 == 0 else hhm_text.random_columns(2000 + k, L)
        base_txt.append(hhm_text.hhm_text("@NAME@", f, k))
    names = ["a%06d" % k for k in range(n)]
    texts = [base_txt[k % uniq].replace(b"@NAME@", names[k].encode()) for k in range(n)]
    out = {"templates": n, "L": L, "host_threads": threads}

    def one(binary, tag, sidecar):
        cmd = [os.path.join(BIN, binary), "-i", qpath, "-d", base, "-nocontxt", "-premerge", "0", "-cpu", str(threads), "-o",
               os.path.join(tmp, tag + ".hhr"), "-scores", os.path.join(tmp, tag + ".scores"), "-v", "0"]
        env = dict(os.environ, HHV_DROPIN_TIMING="1", HHV_SIDECAR=sidecar)
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, env=env, timeout=600)
        dt = time.time() - t0
        assert r.returncode == 0, r.stderr.decode()[-1500:]
        phases = None
        for line in r.stderr.decode().splitlines():
            m = re.search(r"read ([\d.]+) s, upload ([\d.]+) s, device prepare ([\d.]+) s, masks ([\d.]+) s, align\+hits ([\d.]+) s, paths\+Hit ([\d.]+) s, other ([\d.]+) s", line)
            if m and "hhviterbirunner_hip" in line:
                v = [float(x) for x in m.groups()]
                phases = {"alignment_call_s": round(sum(v), 4), "read_s": v[0], "align_and_hits_s": v[4]}
                break
        scores = sorted(l for l in open(os.path.join(tmp, tag + ".scores")).read().splitlines() if not l.startswith(("Date", "Command", "FILE", "COMM")))
        return round(dt, 3), phases, scores

    with tempfile.TemporaryDirectory() as tmp:
        base, qpath = build_db(tmp, query, texts, names, 9)
        t_ref, _, s_ref = one("hhsearch_cpu", "ref", "0")
        t_cold, ph_cold, s_cold = one("hhsearch_hip", "cold", "0")
        t_w, ph_w, s_w = one("hhsearch_hip", "write", "1")
        t_s, ph_s, s_s = one("hhsearch_hip", "side", "1")
        out.update({"reference_process_s": t_ref, "dropin_cold_process_s": t_cold, "dropin_cold_writing_sidecar_process_s": t_w,
                    "dropin_cold_with_sidecar_process_s": t_s,
                    "dropin_cold_alignment": ph_cold, "dropin_cold_with_sidecar_alignment": ph_s,
                    "sidecar_bytes": os.path.getsize(base + "_hhm.ffdata.hhvside"),
                    "same_scores_file": s_ref == s_cold == s_w == s_s})
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    L = 300
    rng = np.random.default_rng(5)
    qfs = [hhm_text.random_columns(900 + k, L) for k in range(nq)]
    queries = [hhm_text.hhm_text("query%02d" % k, qfs[k], 900 + k) for k in range(nq)]
    uniq = min(n, 400)
    base_txt = []
    for k in range(uniq):
        f = hhm_text.mutate_columns(k, qfs[k % nq], 0.4) if k % 20 == 0 else hhm_text.random_columns(2000 + k, L)
        base_txt.append(hhm_text.hhm_text("@NAME@", f, k))
    names = ["a%06d" % k for k in range(n)]
    texts = [base_txt[k % uniq].replace(b"@NAME@", names[k].encode()) for k in range(n)]
    out = {"n_templates": n, "L": L, "threads": threads, "n_queries": nq}
    with tempfile.TemporaryDirectory() as tmp:
        base, qpath = build_db(tmp, queries[0], texts, names, 9)
        common = ["-d", base, "-nocontxt", "-premerge", "0"]
        args = ["-i", qpath, "-cpu", str(threads)] + common
        t_cpu, o_cpu = timed(run_app, "hhsearch_cpu", args, os.path.join(tmp, "c1"))
        # one cold process each: without the binary sidecar (every template parsed from the .hhm text), the first run that
        # writes it, and a run that finds it (hh-suite_amd/dropin/hhv_sidecar.h)
        os.environ["HHV_SIDECAR"] = "0"
        t_hip, o_hip = timed(run_app, "hhsearch_hip", args, os.path.join(tmp, "h1"))
        os.environ["HHV_SIDECAR"] = "1"
        t_w, o_w = timed(run_app, "hhsearch_hip", args, os.path.join(tmp, "h1w"))
        t_s, o_s = timed(run_app, "hhsearch_hip", args, os.path.join(tmp, "h1s"))
        # with several threads the reference adds equal-score hits in thread order: compare the sorted hit lines
        same = sorted(o_cpu["scores"]) == sorted(o_hip["scores"]) == sorted(o_w["scores"]) == sorted(o_s["scores"])
        out["hhsearch_one_query"] = {"reference_s": round(t_cpu, 2), "replaced_units_s": round(t_hip, 2),
                                     "replaced_units_writing_sidecar_s": round(t_w, 2), "replaced_units_with_sidecar_s": round(t_s, 2),
                                     "sidecar_bytes": os.path.getsize(base + "_hhm.ffdata.hhvside"), "same_scores_file": same}
        os.environ["HHV_SIDECAR"] = "0"
        qbase = os.path.join(tmp, "queries")
        write_ffindex(qbase, [("q%02d" % k, q.rstrip(b"\n")) for k, q in enumerate(queries)])
        args = ["-i", qbase, "-cpu", str(min(threads, nq))] + common
        t_cpu, o_cpu = timed(run_omp, "hhsearch_omp_cpu", args, os.path.join(tmp, "c2"))
        t_hip, o_hip = timed(run_omp, "hhsearch_omp_hip", args, os.path.join(tmp, "h2"))
        same = all(o_cpu[k] == o_hip[k] for k in o_cpu) and sorted(o_cpu) == sorted(o_hip)
        out["hhsearch_omp_%d_queries" % nq] = {"reference_s": round(t_cpu, 2), "replaced_units_s": round(t_hip, 2),
                                                "same_hhr_files": same}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
