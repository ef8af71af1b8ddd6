"""CPU: the C-ABI shared library loads and exports every symbol include/vneti.h declares."""
import ctypes
import os

import pytest

from view_neti_amd import lib


@pytest.mark.parametrize("precision,code", [("fp16", 0), ("bf16", 1)])
def test_header_symbols_exported(precision, code):
    """both builds of the library (fp16: libvneti_hip.so, bf16: libvneti_hip_bf16.so, the same sources with -DVN_BF16)"""
    path = lib.so_path(precision)
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    so = ctypes.CDLL(path)
    names = lib.declared_symbols()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, f"declared in vneti.h but not exported: {missing}"
    assert so.vneti_version() == 1 and so.vneti_precision() == code


def test_one_precision_per_process():
    lib.load()
    with pytest.raises(RuntimeError):
        lib.set_precision("bf16" if lib.precision() == "fp16" else "fp16")
    lib.set_precision(lib.precision())  # re-stating the loaded one is fine


def test_error_convention_no_gpu():
    """argument validation happens before any HIP call, so it is testable without a GPU."""
    l = lib.load()
    d = lib.GemmDesc()
    rc = l.vneti_gemm_f16(ctypes.byref(d), None)
    assert rc < 0
    assert "null" in lib.last_error().lower()
    with pytest.raises(RuntimeError):
        lib.check(rc, "gemm")


def test_signatures_cover_header():
    names = set(lib.declared_symbols())
    covered = {"vneti_" + k for k in lib.SIGNATURES} | {"vneti_version", "vneti_precision", "vneti_last_error", "vneti_gemm_f16",
                                                       "vneti_groupnorm_ws_floats"} | {"vneti_" + k for k in lib.LL_FUNCS} | {"vneti_" + k for k in lib.INT_FUNCS}
    assert names <= covered, f"no ctypes signature for {sorted(names - covered)}"
