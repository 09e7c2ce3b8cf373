"""Helpers shared by tests/golden/make_real_fixture.py and tests/test_real_profile.py."""
import os

import numpy as np

WINDOWS = [(1, 431), (40, 260), (200, 431), (300, 360)]
HERE = os.path.dirname(os.path.abspath(__file__))


def window(p, tr, a, b):
    """Prepared sub-HMM of template columns a..b: column k of the window = column a+k-1, tr row 0 = tr[a-1]."""
    return (np.ascontiguousarray(np.vstack([np.zeros((1, 20), np.float32), p[a:b + 1]])),
            np.ascontiguousarray(tr[a - 1:b + 1]))


def load_prepared():
    z = np.load(os.path.join(HERE, "golden", "query_hhm_prepared.npz"))
    return z["qp"], z["qtr"], z["tp"], z["ttr"]
