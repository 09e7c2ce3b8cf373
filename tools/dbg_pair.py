import os, sys
sys.path.insert(0, "hh-suite_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
import numpy as np
from pyhhv import capi, synth
from pyoracle import Oracle, make_params
orc = Oracle()
Lq, local = 321, 0
par = make_params(local=local)
qf, qtr = synth.make_query(1, Lq)
qf2, qtr2 = synth.make_query(77, Lq)
tp, ttr = synth.make_homolog(2, qf, L=200)
c = capi.Context(local=local)
ts = c.upload([tp], [ttr])
os.environ["HHV_PAIR"] = "0"
c.set_query(qf2, qtr2); r0 = c.align(ts).copy(); print("two launches, other query:", r0)
os.environ["HHV_PAIR"] = "1"
c.set_query(qf, qtr); r1 = c.align(ts).copy(); print("pair, real query        :", r1, "kernel ms", c.last_kernel_ms())
os.environ["HHV_PAIR"] = "0"
r2 = c.align(ts).copy(); print("two launches, real query:", r2, "kernel ms", c.last_kernel_ms())
a = orc.align(par, qf, qtr, tp, ttr, want_bt=False); print("oracle", a.score, a.i2, a.j2)
os.environ["HHV_PAIR"] = "1"
r3 = c.align(ts).copy(); print("pair again              :", r3)
