"""Case list shared by tests/golden/make_golden.py (reference side) and tests/test_golden.py."""
import numpy as np

from common import workload
from pyoracle import make_params

# name, seed, Lq, n templates, Lt range, local, egq, egt, celloff round (mask from a first-round path)
CASES = [
    {"name": "global_small", "seed": 1, "Lq": 37, "n": 4, "lo": 5, "hi": 60, "local": 0, "egq": 0.0, "egt": 0.0, "celloff": 0},
    {"name": "local_small", "seed": 2, "Lq": 64, "n": 4, "lo": 20, "hi": 90, "local": 1, "egq": 0.0, "egt": 0.0, "celloff": 0},
    {"name": "global_endgaps", "seed": 3, "Lq": 65, "n": 3, "lo": 30, "hi": 120, "local": 0, "egq": 0.3, "egt": 0.1, "celloff": 0},
    {"name": "local_300", "seed": 4, "Lq": 300, "n": 3, "lo": 250, "hi": 320, "local": 1, "egq": 0.0, "egt": 0.0, "celloff": 0},
    {"name": "global_300", "seed": 5, "Lq": 300, "n": 3, "lo": 300, "hi": 300, "local": 0, "egq": 0.0, "egt": 0.0, "celloff": 0},
    {"name": "local_celloff", "seed": 6, "Lq": 120, "n": 3, "lo": 80, "hi": 150, "local": 1, "egq": 0.0, "egt": 0.0, "celloff": 1},
    {"name": "global_celloff", "seed": 7, "Lq": 120, "n": 3, "lo": 80, "hi": 150, "local": 0, "egq": 0.0, "egt": 0.0, "celloff": 1},
    {"name": "query_431", "seed": 8, "Lq": 431, "n": 2, "lo": 150, "hi": 250, "local": 1, "egq": 0.0, "egt": 0.0, "celloff": 0},
    {"name": "tiny", "seed": 9, "Lq": 1, "n": 3, "lo": 1, "hi": 3, "local": 0, "egq": 0.0, "egt": 0.0, "celloff": 0},
]


def build_case(spec):
    """-> (par, qf, qtr, tps, ttrs, masks|None).  Cell-off masks are derived with the ORACLE's
    ExcludeAlignment from the oracle's first-round path (both pinned to the reference elsewhere), so
    that this function needs no reference library."""
    par = make_params(local=spec["local"], egq=spec["egq"], egt=spec["egt"])
    qf, qtr, tps, ttrs = workload(spec["seed"], spec["Lq"], spec["n"], spec["lo"], spec["hi"],
                                  homolog_every=1 if spec["celloff"] else 2)
    masks = None
    if spec["celloff"]:
        from pyoracle import Oracle
        o = Oracle()
        masks = []
        for p, tr in zip(tps, ttrs):
            a = o.align(par, qf, qtr, p, tr, want_path=True)
            masks.append(o.exclude_alignment(spec["Lq"], p.shape[0] - 1, a.i_steps, a.j_steps, a.nsteps))
    return par, qf, qtr, tps, ttrs, masks
