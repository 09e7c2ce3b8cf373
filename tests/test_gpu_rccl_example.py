"""The sharded search as a native multi-process program (examples/sharded_search_rccl.cpp: the C ABI + librccl, no torch):
hhv_shard_plan -> per-rank hhv_topk(d_out) -> ncclAllGather on hhv_stream(ctx) -> hhv_merge_hits.  World 1 runs the whole
RCCL branch on the test box's single GPU and must equal the plain top-K of one search over all templates (--check: record
for record); world 2 runs as two processes on two GPUs when the box has them (RCCL refuses two ranks on one device)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "sharded_search_rccl")


def run(*args, timeout=300):
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-C", ROOT, "example_rccl"])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=timeout, env=env)
    return r.returncode, r.stdout + r.stderr


def n_gpus():
    from pyhhv import capi
    return capi.device_count()


@pytest.mark.parametrize("extra", [(), ("--backtrace",), ("--zipf", "--lq", "120")])
def test_one_rank_through_rccl_equals_the_plain_search(extra):
    rc, out = run("--world", 1, "--templates", 3000, "--lt", 120, "--lq", 150, "--topk", 200, "--steps", 2, "--check", *extra)
    assert rc == 0, out
    assert "identical on all 1 ranks" in out and "OK, identical records" in out, out


def test_two_ranks_on_two_gpus_equal_one_gpu():
    if n_gpus() < 2:
        pytest.skip("one GPU visible: RCCL does not put two ranks on one device")
    for extra in ((), ("--backtrace", "--zipf")):
        rc, out = run("--world", 2, "--templates", 6000, "--lt", 150, "--lq", 200, "--topk", 300, "--steps", 2, "--check", *extra)
        assert rc == 0, out
        assert "identical on all 2 ranks" in out and "OK, identical records" in out, out


def test_more_ranks_than_gpus_is_refused_with_a_message():
    rc, out = run("--world", n_gpus() + 1, "--templates", 100)
    assert rc == 7 and "GPUs" in out, out
