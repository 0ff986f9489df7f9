"""CPU: the C-ABI library builds for gfx950, loads, and exports every entry point include/gwdepth.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from gw_depth_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_entry_points():
    text = open(os.path.join(ROOT, "include", "gwdepth.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gwd_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    names = declared_entry_points()
    assert len(names) >= 20
    assert sorted(hip.ENTRY_POINTS) == names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared_entry_points():
        assert hasattr(lib, name), name
    lib.gwd_arch.restype = ctypes.c_char_p
    assert lib.gwd_version() == 10 and lib.gwd_arch() == b"gfx950"


def test_code_object_targets_gfx950_only():
    data = open(hip.LIB_PATH, "rb").read()
    assert b"gfx950" in data and b"gfx942" not in data and b"gfx90a" not in data


def test_compute_entry_points_refuse_cpu_tensors():
    import torch
    hip.set_library(None)
    lib = hip.library()
    x = torch.zeros(4, 8)
    with pytest.raises(hip.HipUnavailable):
        lib.softmax_forward(x, torch.empty_like(x), 4, 8)


def test_workspace_query_needs_no_gpu():
    lib = ctypes.CDLL(hip.LIB_PATH)
    lib.gwd_query_workspace.restype = ctypes.c_int64
    lib.gwd_query_workspace.argtypes = [ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.c_int32]
    d3 = (ctypes.c_int64 * 3)(8, 32, 16)
    assert lib.gwd_query_workspace(0, d3, 3) == 8 * 32 * 16 * 2 * 4
    d4 = (ctypes.c_int64 * 4)(8, 120, 10, 160)
    assert lib.gwd_query_workspace(1, d4, 4) == 8 * 120 * 10 * 160 * 4
    assert lib.gwd_query_workspace(1, d3, 3) == -1 and lib.gwd_query_workspace(7, d3, 3) == -1
    d2 = (ctypes.c_int64 * 2)(8, 480 * 640)                                   # GWD_WS_EVAL: 150 partial records per image
    assert lib.gwd_query_workspace(2, d2, 2) == 8 * 150 * (10 * 8 + 4 * 8)
    d2 = (ctypes.c_int64 * 2)(28, 480 * 640)                                  # GWD_WS_PLANE: 128 partial records per plane
    assert lib.gwd_query_workspace(3, d2, 2) == 28 * 128 * 5 * 8 and lib.gwd_query_workspace(3, (ctypes.c_int64 * 2)(65, 100), 2) == -1
