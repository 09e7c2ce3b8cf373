"""C-ABI surface: the library loads on a CPU-only box, exports every declared symbol and refuses
to compute without a device (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    from pyhhv import capi
    lib = capi.load()
    hdr = open(os.path.join(ROOT, "include", "hhviterbi_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*|int32_t|int64_t|void\*)\s+(hhv_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), "symbol %s declared in include/hhviterbi_hip.h is not exported" % name
    assert set(capi.ABI_SYMBOLS) == declared
    assert lib.hhv_abi_version() == 1
    assert lib.hhv_record_bytes() == 112


def test_struct_sizes():
    from pyhhv import capi
    assert capi.RESULT_DTYPE.itemsize == 16
    assert capi.HIT_DTYPE.itemsize == 40


def test_no_cpu_fallback():
    """Without a GPU hhv_create must fail with HHV_E_DEVICE; with one it must succeed."""
    import torch
    from pyhhv import capi
    if torch.cuda.is_available():
        c = capi.Context()
        c.close()
    else:
        with pytest.raises(capi.HhvError) as e:
            capi.Context()
        assert "no CPU path" in str(e.value)


def test_product_does_not_reference_oracle():
    """The product sources must not include, link or load anything under oracle/."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "hh-suite_amd")):
        for f in files:
            if f.endswith((".h", ".hip", ".cpp", ".py")):
                s = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r"(#include|import|CDLL|dlopen)[^\n]*(oracle|hhv_oracle|libhhref|liboracle)", s):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_arg_errors_do_not_exit():
    from pyhhv import capi
    lib = capi.load()
    assert lib.hhv_pack_profile(None, None, 0, 0, None) == -1
    assert b"hhv_pack_profile" in lib.hhv_last_error()
    assert lib.hhv_align(None, None, 0, None) == -1
    assert lib.hhv_tset_size(None) == 0
