#!/usr/bin/env python3
"""Round-5 question (review item 2d): the multi-bin discriminator's (3 x 3) / stride-(1, 2) layers run as F(5, 3) along the bins
around a (1 x 3) / stride-2 row convolution - 3 column products per output.  Splitting the columns by parity makes the odd phase a
2-tap stride-1 convolution, F(4, 2) of which costs 5 products per 4 outputs; with the even phase's single plain tap that is 2.25
column products per output instead of 3.  Does it pay on the f32 engine?

Measured here, per layer shape of config 2 and alone on the GPU: (a) the GEMM the layer runs today (the batch-grouped (1 x 3) /
stride-2 convolution over the 7 row positions), forward / input gradient / weight gradient; (b) the GEMM the column form would
run instead: per row position and band one 1x1 product over 5 Winograd positions x Q/4 tiles + Q plain outputs = 2.25 Q rows of
C -> K (the same kernels, dense 1x1 instantiation); (c) the bytes the column form adds around it, priced at the rate the existing
row transforms reach (mq_wino1d_input_transform / _output_transform timed on the same tensors): with the column transforms FUSED
into the row transforms (the cheapest possible construction) the GEMM writes 2.25 / 1 as much (M2 instead of M) and the output
transform reads it; the input side grows by 1/8.
usage: python tools/mb_stride2_micro.py"""
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


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def conv_times(d, x, wk, wc, dy):
    y = torch.empty(d.N, d.P, d.Q, d.K, device=DEV)
    dx = torch.empty_like(x)
    dw = torch.empty_like(wk)
    nb = L().mq_conv_wgrad_workspace(C.byref(d))
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    f = timeit(lambda: check(L().mq_conv_fprop(C.byref(d), fptr(x), fptr(wk), None, fptr(y), stream()), "fprop"))
    kf = L().mq_last_kernel().decode()
    b = timeit(lambda: check(L().mq_conv_dgrad(C.byref(d), fptr(dy), fptr(wc), fptr(dx), stream()), "dgrad"))
    kb = L().mq_last_kernel().decode()
    w = timeit(lambda: check(L().mq_conv_wgrad(C.byref(d), fptr(x), fptr(dy), fptr(dw), ptr(ws), nb, stream()), "wgrad"))
    kw = L().mq_last_kernel().decode()
    return (f, b, w), (kf, kb, kw)


# (images per band x bands, W, C, K): the three stride-(1, 2) layers of config 2's band discriminators, D-step batch (real + fake)
for N, W, Cc, K in ((512, 512, 128, 256), (512, 256, 256, 256), (512, 128, 256, 384)):
    G, H, m = 8, 10, 5
    th, npos = H // m, m + 2
    Q = W // 2
    rows = npos * N * th                                   # one-row images of the row form's GEMM
    # (a) today: (1 x 3) / stride 2 over `rows` images of width W, npos * G batch groups
    d = ops.conv_desc(rows, 1, W, Cc, K, 1, 3, stride=(1, 2), pad=(0, 1), groups=npos * G, precision="f32")
    V = torch.randn(rows, 1, W, Cc, device=DEV)
    wk = torch.randn(npos * G * K, 3, Cc, device=DEV) * 0.05
    wc = torch.randn(npos * G * Cc, 3, K, device=DEV) * 0.05
    Z = torch.randn(rows, 1, Q, K, device=DEV)
    (f0, b0, w0), k0 = conv_times(d, V, wk, wc, Z)
    # (b) the column form's GEMM: 1x1, 2.25 Q rows per image
    Q2 = 5 * (Q // 4) + Q
    d2 = ops.conv_desc(rows, 1, Q2, Cc, K, 1, 1, groups=npos * G, precision="f32")
    V2 = torch.randn(rows, 1, Q2, Cc, device=DEV)
    wk2 = torch.randn(npos * G * K, 1, Cc, device=DEV) * 0.05
    wc2 = torch.randn(npos * G * Cc, 1, K, device=DEV) * 0.05
    Z2 = torch.randn(rows, 1, Q2, K, device=DEV)
    (f1, b1, w1), k1 = conv_times(d2, V2, wk2, wc2, Z2)
    # (c) the rate of the existing row transforms on this layer's tensors
    x = torch.randn(N, H, W, Cc, device=DEV)
    Vr = torch.empty(rows, 1, W, Cc, device=DEV)
    ti = timeit(lambda: check(VL().mq_wino1d_input_transform(fptr(x), N, H, W, Cc, m, fptr(Vr), stream()), "wino1d_input"))
    gb_i = 4.0 * (x.numel() + Vr.numel()) / 1e9
    yb = torch.empty(N, H, Q, K, device=DEV)
    bias = torch.zeros(G * K, device=DEV)
    M = torch.randn(rows, 1, Q, K, device=DEV)
    to = timeit(lambda: check(VL().mq_wino1d_output_transform(fptr(M), fptr(bias), N, H, Q, K, G, m, 2, 0.2, None, 1, 0, 1, fptr(yb), stream()),
                              "wino1d_output"))
    gb_o = 4.0 * (M.numel() + yb.numel()) / 1e9
    rate_i, rate_o = gb_i / ti, gb_o / to                  # GB/ms = TB/s
    # extra bytes of the FUSED column form: forward  V2 - V written and read (+1/8 V each), M2 - M written and read (+1.25 M each)
    extra_f = 4.0 * (2 * 0.125 * V.numel() + 2 * 1.25 * Z.numel()) / 1e9
    t_extra = extra_f / min(rate_i, rate_o)
    print(f"N {N} x 10 x {W}  C {Cc} -> K {K}, stride (1, 2):", flush=True)
    print(f"   today  (1x3 s2)  fprop {f0:6.3f}  dgrad {b0:6.3f}  wgrad {w0:6.3f} ms   [{k0[0]} | {k0[1]} | {k0[2]}]")
    print(f"   column (1x1 x2.25 Q) fprop {f1:6.3f}  dgrad {b1:6.3f}  wgrad {w1:6.3f} ms   [{k1[0]} | {k1[1]} | {k1[2]}]")
    print(f"   row transforms today: input {ti:6.3f} ms ({rate_i:4.2f} TB/s), output {to:6.3f} ms ({rate_o:4.2f} TB/s); the fused column form moves "
          f"{extra_f:5.2f} GB more per pass = {t_extra:6.3f} ms")
    print(f"   net per pass: fprop {f0 - f1 - t_extra:+6.3f}  dgrad {b0 - b1 - t_extra:+6.3f}  wgrad {w0 - w1 - 0.5 * t_extra:+6.3f} ms  (positive = the column form wins)",
          flush=True)
