"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/geodiff_hip.h declares; argument validation works without touching a GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from geodiffuser_amd.build import build
    from geodiffuser_amd import _lib
    build()
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "geodiff_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gd_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound(lib):
    from geodiffuser_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(syms)
    assert lib.gd_version() == 3


def test_argument_validation_without_gpu(lib):
    # null pointers / unsupported sizes are rejected before any HIP call
    assert lib.gd_ddim_step(None, None, None, 1.0, 0.5, 0.5, None, 10, 2, None) == -1
    assert b"null" in lib.gd_last_error()
    assert lib.gd_attn_fwd(None, 1, 64, 64, 64, 0.125, 0, None) == -1
    from geodiffuser_amd._lib import GdAttnSeg
    seg = (GdAttnSeg * 1)(GdAttnSeg(1, 1, 1, 1, 0, 1, 0))
    assert lib.gd_attn_fwd(seg, 1, 64, 64, 40, 0.125, 0, None) == -4          # head dim 40 unsupported
    assert b"head dim" in lib.gd_last_error()
    assert lib.gd_error_string(-4) == b"unsupported configuration"
    assert lib.gd_rasterize_workspace_bytes(4096, 64, ctypes.c_float(1.3 / 64 * 2)) > 4096 * 4


def test_ops_refuse_cpu_tensors():
    import torch
    from geodiffuser_amd import ops, GeodiffError
    with pytest.raises(GeodiffError):
        ops.ddim_step(torch.zeros(4), torch.zeros(4), None, 1.0, 0.5, 0.5)


def test_missing_library_fails_loudly(tmp_path):
    from geodiffuser_amd import _lib
    saved = _lib._lib
    _lib._lib = None
    try:
        with pytest.raises(_lib.GeodiffError):
            _lib.load(str(tmp_path / "nope.so"))
    finally:
        _lib._lib = saved
