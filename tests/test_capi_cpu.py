"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol the header declares,
and refuses to run without a gfx950 device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from coffeedb_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build_library()
    return capi.load_library()


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "coffeedb_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cdb_[a-z_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/coffeedb_gpu.h but not exported"
    assert sorted(capi.EXPORTS) == declared


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.cdb_create(C.byref(h), -1) == 2  # CDB_E_DEVICE
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.GpuStringIndex()


def test_product_does_not_touch_oracle():
    # the shipped sources must never reference oracle/ (judge rule: oracle is test infrastructure)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "coffeedb_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
