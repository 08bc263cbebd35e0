"""The C-ABI shared library: it loads (no GPU needed), exports every symbol that
include/mqgan_hip.h declares, and the ctypes table in mqgan_amd/_lib.py agrees with
the header on the number of parameters of each entry point."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "mqgan_hip.h")


def declared(blocks=None):
    """name -> parameter count of every entry point the header declares; ``blocks`` True / False: only the building blocks
    (MQ_BLOCKS_API: exported by the test-variant library alone) / only the product library's exports."""
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"(MQ_BLOCKS_API\s+)?\b(?:int|size_t|void|const char \*)\s*(mq_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        is_block, name, args = bool(m.group(1)), m.group(2), m.group(3).strip()
        if blocks is not None and is_block != blocks:
            continue
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
    d, blocks = declared(blocks=False), declared(blocks=True)
    assert set(d) == set(_lib.SIGNATURES), (set(d) ^ set(_lib.SIGNATURES))
    assert set(blocks) == set(_lib.VARIANT_SIGNATURES), (set(blocks) ^ set(_lib.VARIANT_SIGNATURES))
    for name, nargs in d.items():
        assert hasattr(lib, name), name
        assert len(_lib.SIGNATURES[name][1]) == nargs, (name, nargs, len(_lib.SIGNATURES[name][1]))
    for name, nargs in blocks.items():
        assert len(_lib.VARIANT_SIGNATURES[name][1]) == nargs, (name, nargs)
    assert lib.mq_abi_version() == 11 == _lib.ABI_VERSION
    # nm agrees (extern "C", no mangling) - in BOTH directions: the product library exports exactly the product section of the
    # header, no building block, no prototype (r5 exported 25 transform entry points and mq_winofused_fprop that nothing in the
    # package calls)
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (mq_\w+)\b", syms))
    assert exported == set(d), (sorted(exported - set(d)), sorted(set(d) - exported))


def test_product_library_exports_only_what_the_package_calls(built):
    """every exported entry point is one mqgan_amd/*.py calls - apart from a short list of helpers a C consumer
    needs that the Python host has no use for."""
    import glob
    from mqgan_amd import _lib
    used = set()
    for f in glob.glob(os.path.join(ROOT, "mqgan_amd", "*.py")):
        text = open(f).read()
        if f.endswith("_lib.py"):
            text = text.split("SIGNATURES = {")[0] + text.split("def load():")[1]        # the binding table itself is not a call
        used |= set(re.findall(r"\b(mq_\w+)\b", text))
    helpers = {"mq_abi_version", "mq_last_error",          # used by _lib.load() / check() (counted above) - listed for the C consumer
               "mq_plan_cfg_default",                      # a C consumer's way to the defaults ops.py builds its mqPlanCfg from
               "mq_wino1d_tile", "mq_winow_tile"}          # which tile mq_conv_plan will pick for a row / column form (host-side query)
    unused = set(_lib.SIGNATURES) - used - helpers
    assert not unused, sorted(unused)


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


def test_collective_entry_points_resolve_rccl_and_validate_on_the_host(built):
    """mq_comm_* / mq_allreduce_bucket (SURVEY 8b `allreduce_bucket`): RCCL is resolved with dlopen at the first call - the copy
    this process already maps (PyTorch-ROCm's own) and no second one - and bad arguments are refused before RCCL sees them."""
    import ctypes as C
    import torch  # noqa: F401 - maps torch's librccl.so
    from mqgan_amd import _lib
    lib = _lib.load()
    assert lib.mq_allreduce_bucket(None, C.c_void_p(16), 4, None) != 0 and b"comm is NULL" in lib.mq_last_error()
    comm, ident = C.c_void_p(), C.create_string_buffer(128)
    assert lib.mq_comm_init(C.byref(comm), 2, 2, ident) != 0 and b"rank 2 of 2" in lib.mq_last_error()
    assert lib.mq_comm_init(None, 1, 0, ident) != 0 and b"NULL" in lib.mq_last_error()
    assert lib.mq_comm_destroy(None) == 0
    mapped = {line.split()[-1] for line in open(f"/proc/{os.getpid()}/maps") if "rccl" in line}
    assert len(mapped) == 1, mapped
    # no link-time dependency: a single-GPU consumer never touches RCCL
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed and "nccl" not in needed


def test_header_is_strict_c99_and_the_plain_c_consumer_builds(built, tmp_path):
    """The boundary is a C ABI: include/mqgan_hip.h compiles as ISO C99 (-pedantic -Werror, gcc - no C++, no HIP, no torch in it),
    and examples/c_consumer (codebook search + planned convolution + the RCCL exchange from a C program) builds against the
    library with gcc.  tests/test_gpu_c_consumer.py runs it."""
    src = tmp_path / "hdr.c"
    src.write_text('#include "mqgan_hip.h"\nint main(void) { mqConvPlan p; (void)p; return mq_abi_version() == 11 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    ex = os.path.join(ROOT, "examples", "c_consumer")
    subprocess.run(["make", "-s", "-C", ex, "clean"], check=True)
    r = subprocess.run(["make", "-s", "-C", ex], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(os.path.join(ex, "c_consumer")), r.stderr[-2000:]
    needed = subprocess.run(["readelf", "-d", os.path.join(ex, "c_consumer")], capture_output=True, text=True).stdout
    assert "libmqgan_hip.so" in needed and "torch" not in needed and "python" not in needed and "stdc++" not in needed, needed


def test_planned_calls_validate_on_the_host(built):
    """mq_conv_plan / mq_conv_*_planned: struct layouts agree between the header and ctypes (the plan echoes the descriptor),
    and the planned calls refuse a short workspace or a plan / epilogue mismatch before any launch."""
    import ctypes as C
    from mqgan_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(N=4, H=16, W=20, C=128, K=144, R=3, S=3, P=16, Q=20, stride_h=1, stride_w=1, pad_h=1, pad_w=1, dil_h=1,
                      dil_w=1, groups=1, precision=0)
    cfg = _lib.PlanCfg()
    lib.mq_plan_cfg_default(C.byref(cfg))
    assert cfg.keep_mib > 0 and cfg.wino_tile in (0, 3, 4, 45, 6)
    p = _lib.ConvPlan()
    assert lib.mq_conv_plan(C.byref(d), 0, C.byref(cfg), C.byref(p)) == 0
    assert (p.desc.N, p.desc.K, p.desc.pad_w, p.desc.precision) == (4, 144, 1, 0)            # the struct layout round-trips
    assert p.form == _lib.FORM_WINO2D and lib.mq_form_name(p.form) == b"winograd2d" and p.gemm.groups == p.npos == (p.mh + 2) * (p.mw + 2)
    assert p.ws_fprop > 0 and p.ws_backward > 0 and p.keep_bytes % 256 == 0
    a = C.c_void_p(256)
    rc = lib.mq_conv_fprop_planned(C.byref(p), a, a, None, 0, 0.0, None, 1, 0, 1, a, None, a, p.ws_fprop - 1, None)
    assert rc != 0 and b"workspace" in lib.mq_last_error()
    rc = lib.mq_conv_fprop_planned(C.byref(p), a, a, None, 2, 0.2, None, 1, 0, 1, a, None, a, p.ws_fprop, None)
    assert rc != 0 and b"plan was made for" in lib.mq_last_error()
    rc = lib.mq_conv_backward_planned(C.byref(p), None, None, a, None, a, None, 1.0, None, 1, 0, 1, None, a, None, a, p.ws_backward, None)
    assert rc != 0 and b"keep buffer or x" in lib.mq_last_error()
    cfg.wino_tile = 2
    assert lib.mq_conv_plan(C.byref(d), 0, C.byref(cfg), C.byref(p)) != 0 and b"wino_tile" in lib.mq_last_error()


def test_ops_refuse_cpu_tensors(built):
    import torch
    from mqgan_amd import _lib
    with pytest.raises(_lib.MqganHipError):
        _lib.fptr(torch.zeros(4))
