import os, sys
sys.path.insert(0, "hh-suite_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import numpy as np
from pyhhv import capi
from common import workload
from pyoracle import SSInfo, make_params, Oracle
def say(*a):
    print(*a, flush=True)
Lq = int(sys.argv[1]) if len(sys.argv) > 1 else 300
local = int(sys.argv[2]) if len(sys.argv) > 2 else 0
steps = sys.argv[3] if len(sys.argv) > 3 else "so,bt,hits"
rng = np.random.default_rng(5 + Lq)
S73 = rng.normal(0, 1, (8, 4, 11)).astype(np.float32)
S33 = rng.normal(0, 1, (4, 11, 4, 11)).astype(np.float32)
S37 = rng.normal(0, 1, (4, 11, 8)).astype(np.float32)
par = make_params(local=local, ss_mode=2)
n = 9
qf, qtr, tps, ttrs = workload(9 + Lq, Lq, n, 40, 260, homolog_every=2)
c = capi.Context(local=local, ssw=par["ssw"], ss_mode=2)
c.set_query(qf, qtr)
c.set_ss_tables(S73, S33, S37)
q_ss = (rng.integers(0, 4, Lq + 1), rng.integers(0, 11, Lq + 1), rng.integers(0, 8, Lq + 1))
c.set_query_ss(*q_ss)
t_ss = [(rng.integers(0, 4, p.shape[0]), rng.integers(0, 11, p.shape[0]), rng.integers(0, 8, p.shape[0])) for p in tps]
ts = c.upload(tps, ttrs, t_ss)
o = Oracle()
for mode in (4, 2):
    c.set_ss_mode(mode)
    ss = SSInfo(mode, *q_ss, S73, S33, S37)
    want = [o.align(par, qf, qtr, tps[e], ttrs[e], ss=ss, t_ss=t_ss[e], want_path=True) for e in range(n)]
    if "so" in steps:
        say("mode", mode, "score-only launch ...")
        r = c.align(ts)
        say("  ok:", [(np.float32(a.score) == r["score"][e], (a.i2, a.j2) == (r["i2"][e], r["j2"][e])) for e, a in enumerate(want)])
    if "bt" in steps:
        say("mode", mode, "backtrace launch ...")
        r = c.align(ts, backtrace=True)
        say("  ok:", [(np.float32(a.score) == r["score"][e], (a.i2, a.j2) == (r["i2"][e], r["j2"][e])) for e, a in enumerate(want)])
    if "hits" in steps:
        say("mode", mode, "hits ...")
        h = c.hits(ts)
        say("  ok:", [(np.float32(a.hit_score) == h["score"][e], np.float32(a.score_ss) == h["score_ss"][e], a.nsteps == h["nsteps"][e]) for e, a in enumerate(want)])
say("done")
