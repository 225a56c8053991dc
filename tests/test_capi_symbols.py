"""The C-ABI library loads and exports every symbol include/roomnet_hip.h declares.
No compute calls here (no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from roomnet_amd import _capi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "roomnet_hip.h")).read()
    return sorted(set(re.findall(r"^RN_API [^;(]*?\b(rn_[a-z0-9_]+)\s*\(", text, flags=re.M)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(_capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.isfile(_capi.LIB_PATH), "build the library first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    lib.rn_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.rn_version()


def test_struct_layouts_match_header_constants():
    assert ctypes.sizeof(_capi.rn_stage_ms) == 4 + 4 + 4 * _capi.RN_MAX_STAGES + 4 + 4
    assert ctypes.sizeof(_capi.rn_node_info) == _capi.RN_NAME_LEN + 12
    assert ctypes.sizeof(_capi.rn_conv_stage) == 5 * 4 + 4 + 9 * 8   # 5 ints, pad, 9 pointers


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_capi.RoomNetLibraryError):
        _capi.load_library(str(tmp_path / "libnope.so"))


def test_no_product_module_imports_the_oracle():
    pkg = os.path.join(ROOT, "roomnet_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "tf_ops" not in text, f


def test_ab_library_exports_the_same_symbols_and_only_it_takes_the_round2_flag(weights):
    """Round 6: the round-2 comparison kernels (RN_FLAG_PAIR_32X32, rn_stage23.hip) live in libroomnet_hip_ab.so -- the product
    library's objects plus that one -- and the product library answers the flag with RN_E_INVALID before it touches a device."""
    assert os.path.isfile(_capi.AB_LIB_PATH), "csrc/build.sh builds both libraries"
    ab = ctypes.CDLL(_capi.AB_LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(ab, name), name
    assert os.path.getsize(_capi.LIB_PATH) < os.path.getsize(_capi.AB_LIB_PATH)
    from roomnet_amd.graph import build_graph
    with pytest.raises(ValueError, match="libroomnet_hip_ab.so"):
        _capi.Engine(build_graph(6, 224), weights, device=0, dtype="bf16", max_batch=1, pair32=True, lib_path=_capi.LIB_PATH)
