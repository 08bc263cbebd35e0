#!/usr/bin/env python3
"""Polyphase-Winograd 5x5 / stride-2 layer at the shapes of the oracle-step test: forward, input gradient and weight gradient of
the Winograd route and of the direct kernels against float64, relative L2 and the cosine between the error and the reference
(a systematic scale error shows as |cos| ~ 1), for zero-mean and for one-signed data."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from mqgan_amd import ops  # noqa: E402

DEV = "cuda"


def stats(a, b):
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    e = a - b
    return f"rel {float(e.norm() / b.norm()):.2e} cos {float(e @ b / (e.norm() * b.norm() + 1e-300)):+.3f} max/rms {float(e.abs().max() / b.pow(2).mean().sqrt()):.2e}"


for (N, H, W, C, K) in ((2, 80, 32, 256, 256), (2, 40, 16, 256, 384), (3, 24, 40, 128, 128), (32, 80, 256, 256, 256)):
    for kind in ("zero-mean", "positive"):
        torch.manual_seed(1)
        x = torch.randn(N, H, W, C, device=DEV)
        dy = torch.randn(N, H // 2, W // 2, K, device=DEV)
        if kind == "positive":
            x, dy = F.leaky_relu(x, 0.2), dy.abs()
        w = torch.randn(K, C, 5, 5, device=DEV) * 0.02
        b = torch.zeros(K, device=DEV)
        d = ops.conv_desc(N, H, W, C, K, 5, 5, stride=(2, 2), pad=(2, 2), precision="f32")
        big = N * H * W * C > 5e7
        if not big:
            x64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
            w64 = w.double().cpu().requires_grad_(True)
            y64 = F.conv2d(x64, w64, None, stride=2, padding=2)
            (y64 * dy.double().cpu().permute(0, 3, 1, 2)).sum().backward()
            ref = (y64.detach().permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1), w64.grad)
        res = {}
        for leg, minc in (("winograd", 128), ("direct", 0)):
            keep = ops.WINO_S2D_MIN_CHANNELS
            ops.WINO_S2D_MIN_CHANNELS = minc
            try:
                xv, wv = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
                wk, wc = ops.weight_prep(wv, None, True)
                y = ops.conv(xv, wk, wc, b, d) if not ops._wino_s2d_ok(d) else None
                if y is None:       # the route is only wired through conv_leaky: slope 1 = identity
                    y, _ = ops.conv_leaky(xv, wk, wc, b, d, None, (1, 0, 1), slope=1.0, pregate=True, chain=True)
                (y * dy).sum().backward()
                res[leg] = (y.detach(), xv.grad, wv.grad, y.grad_fn.__class__.__name__)
            finally:
                ops.WINO_S2D_MIN_CHANNELS = keep
        print(f"N{N} {H}x{W} C{C}->K{K} {kind}: {res['winograd'][3]} / {res['direct'][3]}")
        for i, name in enumerate(("y", "dx", "dw")):
            if not big:
                print(f"   {name:2s} winograd vs f64: {stats(res['winograd'][i], ref[i])} | direct vs f64: {stats(res['direct'][i], ref[i])}")
            print(f"   {name:2s} winograd vs direct: {stats(res['winograd'][i], res['direct'][i])}")
