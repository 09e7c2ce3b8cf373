#!/usr/bin/env python3
"""Generate tests/golden/viterbi_golden.json from the REFERENCE ITSELF.

Runs in the build container only (needs oracle/_ref/libhhref.so, i.e. /root/reference at build
time).  Inputs are regenerated from integer seeds by hh-suite_amd/pyhhv/synth.py, so the fixture
only stores the seeds/lengths and the reference's outputs:
  ViterbiResult (score bit pattern, i2, j2), sha256 of the backtrace byte matrix, the Backtrace
  path (i_steps, j_steps, states), matched_cols, sha256 of the per-step scores S and the Hit score
  bit pattern after ScoreForBacktrace.
Each template is aligned as a single-length batch (HMMSimd::MapOneHMM), the definition used for all
mixed-length parity (SURVEY.md 8d cfg 5).

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "hh-suite_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from pyoracle import Ref, make_params  # noqa: E402
from golden_cases import CASES, build_case  # noqa: E402


def main():
    ref = Ref()
    out = {"generator": "tests/golden/make_golden.py", "reference": "soedinglab/hh-suite v3.3.0, oracle/_ref build",
           "cases": []}
    for spec in CASES:
        par, qf, qtr, tps, ttrs, masks = build_case(spec)
        entry = {"spec": spec, "templates": []}
        for e, (p, tr) in enumerate(zip(tps, ttrs)):
            co = None if masks is None else [masks[e]]
            o = ref.align_batch(par, qf, qtr, [p], [tr], replicate=True, celloffs=co, want_path=True)[0]
            ns = o.nsteps
            entry["templates"].append({
                "Lt": int(p.shape[0] - 1),
                "score_bits": int(np.float32(o.score).view(np.uint32)),
                "i2": o.i2, "j2": o.j2,
                "bt_sha256": hashlib.sha256(np.ascontiguousarray(o.bt[1:, 1:]).tobytes()).hexdigest(),
                "nsteps": ns, "matched_cols": o.matched_cols,
                "i_steps": [int(x) for x in o.i_steps[1:ns + 1]],
                "j_steps": [int(x) for x in o.j_steps[1:ns + 1]],
                "states": [int(x) for x in o.states[1:ns + 1]],
                "S_sha256": hashlib.sha256(np.ascontiguousarray(o.S[1:ns + 1]).tobytes()).hexdigest(),
                "hit_score_bits": int(np.float32(o.hit_score).view(np.uint32)),
            })
        out["cases"].append(entry)
    path = os.path.join(ROOT, "tests", "golden", "viterbi_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
