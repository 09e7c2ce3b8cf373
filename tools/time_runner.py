import sys, time, numpy as np
import os; sys.path[:0]=[os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','hh-suite_amd')]
from pyhhv import capi, synth
n=4000; Lq=300
qp,qtr=synth.make_query(3,Lq)
tps,ttrs=[],[]
base=[synth.make_homolog(50+k,qp,L=250) if k%4==0 else synth.make_template(50+k,250) for k in range(64)]
for k in range(n):
    p,t=base[k%64]; tps.append(p); ttrs.append(t)
for rep in range(2):
    t0=time.perf_counter()
    hits,*_=capi.runner_alignment(qp,qtr,tps,ttrs,loc=1,altali=4,ssm=0)
    print("runner: %d templates -> %d hits, %.1f ms"%(n,len(hits),(time.perf_counter()-t0)*1e3), np.bincount(hits['irep']))
