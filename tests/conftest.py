import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # as bench.py and pyhhv/capi.py: before torch brings up HIP (hhv_create's comment)
for p in (os.path.join(ROOT, "hh-suite_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch first: the PyTorch-ROCm wheel carries its own HIP runtime; if libhhviterbi_hip.so (linked against /opt/rocm) brings
# up HIP before torch is imported, the process ends up with two runtimes and torch reports "No HIP GPUs are available".
# Imported here, torch's runtime is the one both use (bench.py imports torch first for the same reason).
try:
    import torch  # noqa: F401
except Exception:  # a box without torch can still run the C-ABI tests
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build what is missing (fresh checkout): the HIP library cross-compiles without a GPU; the checkers are
    plain gcc.  On the GPU box the prebuilt .so files travel with the snapshot and nothing is rebuilt."""
    import subprocess
    need = [os.path.join(ROOT, "hh-suite_amd", "lib", "libhhviterbi_hip.so"),
            os.path.join(ROOT, "hh-suite_amd", "lib", "libhhv_runner.so"),
            os.path.join(ROOT, "tests", "emul", "libwave_emul.so"),
            os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.call(["make", "-C", ROOT, "-j8", "lib", "emul"])
        subprocess.call(["make", "-C", ROOT, "oracle"])


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests must FAIL (not skip) on a GPU box if the HIP library is missing; they are only
    # deselected by the driver's `-m "not gpu"` here on the CPU container.
    pass


@pytest.fixture(scope="session")
def oracle():
    from pyoracle import Oracle, have_oracle
    if not have_oracle():
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from pyoracle import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libhhref.so not built (needs /root/reference at build time)")
    return Ref()
