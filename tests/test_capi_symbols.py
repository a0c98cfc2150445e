"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/psmf_hip.h declares (no compute calls).  CPU only."""

import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "psmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(psmf_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from rpsmf_amd import _capi
    from rpsmf_amd.build import build_library

    build_library()
    lib = _capi.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in psmf_hip.h but not exported"
        assert name in _capi.SIGNATURES, f"{name} has no ctypes signature"
    assert sorted(_capi.SIGNATURES) == declared


def test_device_path_fails_loudly_without_gpu():
    from rpsmf_amd import _capi

    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_capi.PsmfError, match="no HIP device"):
        _capi.DeviceFilter(100, 4)
    import numpy as np

    import rpsmf_amd as psmf

    f = psmf.PSMFIter(np.zeros((0, 1)), np.zeros((8, 2)), np.eye(2), np.zeros((2, 1)), np.eye(2), {0: np.eye(2)},
                      {0: 1.0}, psmf.RandomWalk())
    with pytest.raises(_capi.PsmfError):
        f.step({1: np.zeros((8, 1))}, 1, 1)     # default backend is the device: no silent CPU fallback


def test_config_struct_matches_header_field_order():
    from rpsmf_amd import _capi

    text = open(os.path.join(ROOT, "include", "psmf_hip.h")).read()
    body = text[text.index("typedef struct {", text.index("Mode table")):text.index("} psmf_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b(?:int32_t|double)\s+([^;]+);", body)
    names = [n.strip() for grp in fields for n in grp.split(",")]
    assert names == [f[0] for f in _capi.PsmfConfig._fields_]


def test_integration_stub_lists_the_config_fields_in_header_order():
    """INTEGRATION.md shows the ctypes stub a maintainer of the reference would add: its psmf_config must be the header's."""
    from rpsmf_amd import _capi

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = text[text.index("class psmf_config(C.Structure)"):text.index("dp = C.POINTER(C.c_double)")]
    ints = " ".join(re.findall(r'"([a-zA-Z_0-9 ]+)"\s*\)?\.split\(\)', stub)[:1])
    parts = re.findall(r'"([a-zA-Z_0-9 ]+)"', stub)
    names = " ".join(parts).split()
    assert names == [f[0] for f in _capi.PsmfConfig._fields_], (names, ints)
    assert f"abi_version={_capi.ABI_VERSION}" in text
