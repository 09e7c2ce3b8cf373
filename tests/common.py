"""Shared helpers for the parity tests: seeded workloads and comparisons."""
import numpy as np

from pyhhv import synth


def workload(seed, Lq, n, Lt_lo, Lt_hi, homolog_every=2):
    """Query + n templates (every `homolog_every`-th one derived from the query)."""
    rng = np.random.default_rng(seed)
    qf, qtr = synth.make_query(1000 + seed, Lq)
    tps, ttrs = [], []
    for e in range(n):
        Lt = int(rng.integers(Lt_lo, Lt_hi + 1))
        if homolog_every and e % homolog_every == 0:
            p, tr = synth.make_homolog(5000 + seed * 131 + e, qf, L=Lt)
        else:
            p, tr = synth.make_template(7000 + seed * 131 + e, Lt)
        tps.append(p)
        ttrs.append(tr)
    return qf, qtr, tps, ttrs


def same_float(a, b):
    """Equal as IEEE values (+0 == -0); the engine uses v_max_f32 where the reference uses MAXPS,
    which can differ in the sign of an exact zero only."""
    return np.float32(a) == np.float32(b)
