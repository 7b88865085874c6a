"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/snnhip.h declares (no compute calls -- there is no GPU in this container)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(snnhip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built):
    import shadernn_amd as snn
    from shadernn_amd import capi

    path = snn.load_library()
    lib = ctypes.CDLL(path)
    declared = _declared_symbols(os.path.join(ROOT, "include", "snnhip.h"))
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), "libsnnhip.so does not export " + name
    # and the Python binding lists exactly the declared set
    assert sorted(capi.SIGNATURES) == declared


def test_library_is_in_tree_and_has_gfx950_code(built):
    from shadernn_amd import capi

    assert capi.LIB_PATH.startswith(ROOT)
    blob = open(capi.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_errors_are_reported_not_swallowed(built):
    """Without a GPU the context cannot be created: the call must fail loudly (no CPU fallback exists)."""
    import torch

    import shadernn_amd as snn

    snn.load_library()
    if torch.cuda.is_available():
        return
    try:
        snn.Context(0)
    except snn.SnnHipError as e:
        assert e.code in (-1, -2)
    else:
        raise AssertionError("Context(0) succeeded without a GPU")


def test_options_registry_overrides_the_environment(built, monkeypatch):
    """snnhip_set_option / snnhip_get_option (no GPU needed): an override wins over the environment variable of the same name, NULL removes it, names
    outside the SNNHIP_ namespace are rejected."""
    import shadernn_amd as snn

    snn.load_library()
    monkeypatch.delenv("SNNHIP_CONV", raising=False)
    assert snn.get_option("SNNHIP_CONV") is None
    monkeypatch.setenv("SNNHIP_CONV", "generic")
    assert snn.get_option("SNNHIP_CONV") == "generic"
    snn.set_option("SNNHIP_CONV", "mfma")
    try:
        assert snn.get_option("SNNHIP_CONV") == "mfma"
    finally:
        snn.set_option("SNNHIP_CONV", None)
    assert snn.get_option("SNNHIP_CONV") == "generic"
    try:
        snn.set_option("PATH", "x")
    except snn.SnnHipError as e:
        assert e.code == -1
    else:
        raise AssertionError("set_option accepted a name outside SNNHIP_")
