"""CPU-only checks of the drop-in boundary: the shared library loads, exports every symbol that
include/meso_hip.h declares, and refuses to compute without a HIP device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "meso_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(meso_[a-z0-9_]+)\s*\(", src)) - {"meso_host_exchange_fn"})


def test_header_and_loader_agree():
    from meso_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()


def test_library_exports_every_declared_symbol(meso_lib):
    for name in _declared():
        assert hasattr(meso_lib, name), name


def test_version_and_seed_are_host_callable(meso_lib, oracle):
    assert meso_lib.meso_version() >= 100
    M = oracle.meso_lib()
    for seed, step in ((419084618, 0), (419084618, 1), (1, 123456789012), (-5, 77)):
        assert meso_lib.meso_seed_now(seed, step) == M.meso_seed_now(seed, step)


def test_no_cpu_fallback(meso_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = meso_lib.meso_init(0, C.byref(h))
    assert rc != 0 and b"HIP" in meso_lib.meso_last_error()


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "meso_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, f
