#!/usr/bin/env python3
"""Round-5 measurement (review item 2c): F(4x5, 3x3) with the transforms inside the GEMM kernel (csrc/winofused.hip) against the
shipped sequence input transform -> batch-grouped 1x1 GEMM -> output transform (mq_conv_fprop_planned), on the refiner's
K = 64 layers of config 2 (32 images of 512 x 90).  Prints the time of each, the HBM bytes each must move at least, and the error
of both against a float64 convolution on a small sample.  usage: python tools/winofused_micro.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd import _lib as _mqlib  # noqa: E402

ge.build_variants()          # the transform entry points / the prototype live in the test-variant library only (round 6)
VL = _mqlib.load_variants

from mqgan_amd.ops import L, check, fptr, ptr, stream  # noqa: E402

DEV = "cuda"


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for N, H, W, Cc, K in ((32, 512, 90, 64, 64), (32, 512, 90, 192, 64), (32, 256, 90, 64, 64), (2, 24, 35, 64, 64)):
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Cc, device=DEV)
    w = torch.randn(K, Cc, 3, 3, device=DEV) * (1.0 / (3 * Cc ** 0.5))
    bias = torch.randn(K, device=DEV)
    wk, _ = ops.weight_prep(w, None, False)                                   # KRSC, channel-blocked
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, pad=(1, 1), precision="f32")
    p = ops.conv_plan(d, False)
    assert ops.form_name(p) == "winograd2d" and (p.mh, p.mw) == (4, 5), (ops.form_name(p), p.mh, p.mw)
    ws = torch.empty(max(p.ws_fprop, 16), dtype=torch.uint8, device=DEV)
    y0 = torch.empty(N, H, W, K, device=DEV)
    planned = lambda: check(L().mq_conv_fprop_planned(C.byref(p), fptr(x), fptr(wk), fptr(bias), 0, 0.0, None, 1, 0, 1, fptr(y0), None,
                                                      ptr(ws), p.ws_fprop, stream()), "planned")
    U = torch.empty(42 * K * Cc, device=DEV)
    check(VL().mq_winon_weight_transform(fptr(wk), K, Cc, 0, fptr(U), 4, 5, stream()), "weight transform")
    Ut = torch.empty_like(U)
    y1 = torch.empty(N, H, W, K, device=DEV)
    fused = lambda: check(VL().mq_winofused_fprop(fptr(x), fptr(U), fptr(bias), N, H, W, Cc, K, fptr(Ut), fptr(y1), stream()), "fused")
    t0, t1 = timeit(planned), timeit(fused)
    planned(); fused()
    torch.cuda.synchronize()
    rel = float((y1 - y0).norm() / y0.norm())
    tiles = N * ((H + 3) // 4) * ((W + 4) // 5)
    gb_min = 4.0 * (x.numel() + y0.numel()) / 1e9
    gb_seq = 4.0 * (x.numel() + 2 * 42 * tiles * Cc + 2 * 42 * tiles * K + y0.numel()) / 1e9
    msg = f"N {N} {H}x{W} C {Cc} -> K {K}: planned sequence {t0:7.3f} ms ({gb_seq:5.2f} GB: V and M written and read), fused kernel {t1:7.3f} ms " \
          f"({gb_min:5.2f} GB), ratio {t0 / t1:4.2f}; fused vs planned relative L2 {rel:.2e}"
    if N * H * W <= 4096:                                                     # float64 reference on the small case
        ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
        e0 = float((y0.double().cpu() - ref).norm() / ref.norm())
        e1 = float((y1.double().cpu() - ref).norm() / ref.norm())
        msg += f"; against float64: planned {e0:.2e}, fused {e1:.2e}; largest element error of the fused kernel {float((y1.double().cpu() - ref).abs().max() / ref.pow(2).mean().sqrt()):.2e} of the rms"
    print(msg, flush=True)
