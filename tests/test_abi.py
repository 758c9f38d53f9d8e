"""CPU: the C-ABI library loads and exports every symbol include/metalign_hip.h declares (no compute)."""
import os
import re

from metalign_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "metalign_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_list_agree():
    assert declared_symbols() == sorted(_hip.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = _hip.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.mg_abi_version() == 1


def test_record_layout_is_16_bytes():
    assert _hip.REC_DTYPE.itemsize == 16


def test_no_device_means_loud_failure():
    """Without a GPU the product path must raise, never fall back to a CPU implementation."""
    lib = _hip.load_library()
    if lib.mg_device_count() > 0:
        return  # on the GPU box this property is vacuous
    import pytest
    _hip.Hip.reset()
    with pytest.raises(_hip.HipUnavailable):
        _hip.Hip.get(0)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "metalign_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
                assert "libmg_oracle" not in src, f
