"""The C-ABI shared library: it loads (no GPU needed), exports every symbol that
include/mqgan_hip.h declares, and the ctypes table in mqgan_amd/_lib.py agrees with
the header on the number of parameters of each entry point."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mqgan_hip.h")


def declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|size_t|const char \*)\s*(mq_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else args.count(",") + 1
        out[name] = n
    return out


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()


def test_header_parses():
    d = declared()
    assert len(d) > 50 and "mq_conv_fprop" in d and "mq_fsq_forward" in d


def test_library_exports_every_declared_symbol(built):
    from mqgan_amd import _lib
    lib = _lib.load()
    d = declared()
    assert set(d) == set(_lib.SIGNATURES), (set(d) ^ set(_lib.SIGNATURES))
    for name, nargs in d.items():
        assert hasattr(lib, name), name
        assert len(_lib.SIGNATURES[name][1]) == nargs, (name, nargs, len(_lib.SIGNATURES[name][1]))
    assert lib.mq_abi_version() == 7 == _lib.ABI_VERSION
    # nm agrees (extern "C", no mangling)
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for name in d:
        assert re.search(rf"\bT {name}\b", syms), name


def test_host_side_validation_rejects_bad_shapes_without_a_gpu(built):
    """Descriptors are validated on the host before any launch."""
    import ctypes as C
    from mqgan_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(N=3, H=1, W=8, C=4, K=4, R=1, S=3, P=1, Q=8, stride_h=1, stride_w=1, pad_h=0, pad_w=1, dil_h=1,
                      dil_w=1, groups=2)
    rc = lib.mq_conv_fprop(C.byref(d), C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), None)
    assert rc != 0 and b"groups" in lib.mq_last_error()
    assert lib.mq_conv_wgrad_workspace(C.byref(d)) == 0


def test_ops_refuse_cpu_tensors(built):
    import torch
    from mqgan_amd import _lib
    with pytest.raises(_lib.MqganHipError):
        _lib.fptr(torch.zeros(4))
