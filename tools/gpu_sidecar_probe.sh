#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
python - <<'PY'
import os, sys, subprocess, tempfile, time
sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np, hhm_text
from test_dropin_apps import BIN, build_db
n, L = 10000, 300
qf = hhm_text.random_columns(900, L)
query = hhm_text.hhm_text("query00", qf, 900)
uniq = 400
base_txt = []
for k in range(uniq):
    f = hhm_text.mutate_columns(k, qf, 0.4) if k % 20 == 0 else hhm_text.random_columns(2000 + k, L)
    base_txt.append(hhm_text.hhm_text("@NAME@", f, k))
names = ["a%06d" % k for k in range(n)]
texts = [base_txt[k % uniq].replace(b"@NAME@", names[k].encode()) for k in range(n)]
with tempfile.TemporaryDirectory() as tmp:
    base, qpath = build_db(tmp, query, texts, names, 9)
    for threads in (32, 4):
      for mode, env in (("no sidecar", {"HHV_SIDECAR": "0"}), ("writes sidecar", {"HHV_SIDECAR": "1"}), ("reads sidecar", {"HHV_SIDECAR": "1"})):
        e = dict(os.environ, HHV_DROPIN_TIMING="1", **env)
        cmd = [os.path.join(BIN, "hhsearch_hip"), "-i", qpath, "-d", base, "-nocontxt", "-premerge", "0", "-cpu", str(threads), "-o", tmp + "/o.hhr", "-v", "0"]
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, env=e)
        dt = time.time() - t0
        print("threads %d %-15s %.2f s" % (threads, mode, dt))
        for l in r.stderr.decode().splitlines():
            if "hhviterbirunner_hip" in l or "posterior" in l.lower():
                print("   ", l[:400])
      os.remove(base + "_hhm.ffdata.hhvside")
PY
