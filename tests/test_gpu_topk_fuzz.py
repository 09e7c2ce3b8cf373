"""The one-launch top-K of a small set and the one-key-per-thread merge (hhv_topk.hip: topk_small_kernel, merge_hits_small_kernel)
fuzzed against std::sort on the host by a program that includes the kernel source (tools/topk_ubench.hip `fuzz`): random n <= 16 384
and K <= 1024, both branches of the selection (bound from the thread maxima / radix), record and hit sources, ranking keys, global
ids, adversarial orders (all equal, few distinct scores, ascending, descending, the best keys crowded into a few threads, +-FLT_MAX);
merges of up to 4096 records with padding records and K beyond the valid ones.  Order: score descending, ties by the smaller index -
what the reference's caller establishes by sorting the hit list (src/hhhit.h:116-126)."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 20260930])
def test_small_set_topk_and_merge_against_std_sort(seed):
    exe = os.path.join(ROOT, "build", "topk_ubench")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", ROOT, "topk_fuzz"])
    r = subprocess.run([exe, "fuzz", "12", str(seed)], capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and line, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(line[-1])
    assert d["mismatches"] == 0 and d["topk_small_cases"] > 200 and d["merge_cases"] > 200 and d["of_them_radix_branch_forced"] > 50, d
