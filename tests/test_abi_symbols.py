"""The C-ABI library loads without a GPU and exports exactly what include/ebvo_hip.h declares."""
import ctypes
import os
import re

import pytest

from edge_based_visual_odometry_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "ebvo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ebvo_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(_lib.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load_library()
    for name in header_symbols():
        assert hasattr(lib, name), name
    header = open(os.path.join(ROOT, "include", "ebvo_hip.h")).read()
    assert lib.ebvo_abi_version() == int(re.search(r"#define EBVO_ABI_VERSION (\d+)", header).group(1)) == 6
    assert lib.ebvo_strerror(-2) == b"output capacity too small"


def test_argument_errors_need_no_gpu():
    lib = _lib.load_library()
    assert lib.ebvo_ctx_create(0, 0, 0, None) == _lib.EBVO_ERR_ARG
    assert lib.ebvo_epipolar_lines(None, None, 0, None) == _lib.EBVO_ERR_ARG


def test_no_cpu_fallback_without_device():
    """On a machine without a usable HIP device the context cannot be created (the product path
    fails loudly instead of computing on the CPU)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = _lib.load_library()
    ctx = ctypes.c_void_p()
    rc = lib.ebvo_ctx_create(0, 64, 64, ctypes.byref(ctx))
    assert rc == _lib.EBVO_ERR_HIP and not ctx.value
    from edge_based_visual_odometry_amd.api import Context
    with pytest.raises(_lib.EbvoError):
        Context(64, 64)
