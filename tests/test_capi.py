"""The C-ABI library exports every symbol include/jqc_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

from joltqc_amd.backend import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    L.build_library()
    so = ctypes.CDLL(L.LIB_PATH)
    with open(os.path.join(ROOT, "include", "jqc_hip.h")) as f:
        hdr = f.read()
    names = set(re.findall(r"\b(jqc_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 9
    for n in names:
        assert hasattr(so, n), n


def test_compile_only_needs_no_gpu_and_rejects_bad_class():
    import pytest
    h = L.gen_jk_kernel((1, 0, 0, 0), True, True, False, False, L.ALGO_1Q1T, compile_only=True)
    assert h >= 0
    with pytest.raises(RuntimeError):
        L.gen_jk_kernel((0, 1, 0, 0), True, True, False, False, L.ALGO_1Q1T, compile_only=True)
    with pytest.raises(RuntimeError):
        L.gen_jk_kernel((5, 0, 0, 0), True, True, False, False, L.ALGO_1Q1T, compile_only=True)


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        L.require_gpu()


def test_reference_import_name_resolves_to_this_package():
    """``import jqc.pyscf`` (the reference's import line, jqc/pyscf/__init__.py:20,121) gives the same three entry points."""
    import jqc.pyscf as ref_name
    import joltqc_amd.pyscf as own
    assert ref_name.apply is own.apply and ref_name.get_default_config is own.get_default_config and ref_name.reset is own.reset
    assert sorted(ref_name.__all__) == ["apply", "get_default_config", "reset"]
