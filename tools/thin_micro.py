#!/usr/bin/env python3
"""The thin 1x1 layers of config 2 (first / last layers of the refiner and of the two discriminators: 12 ... 28 channels on
one side) as stand-alone launches, timed in the steady state and priced against the bytes they have to move: these layers
are HBM-bound (2 ... 12 flop per byte), so the figure of merit is TB/s, not TF/s.
usage: thin_micro.py [fdw] [seconds-of-warm-up]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mqgan_amd import ops

# (name, N, H, W, C, K, groups): what tools/layer_profile.py prints for the 1x1 layers of a config-2 step
SHAPES = [
    ("mbd_in   24->128 g8", 512, 10, 512, 24, 128, 8),
    ("mbd_in   24->128 g8 (G step)", 256, 10, 512, 24, 128, 8),
    ("patch_in 28->256", 64, 80, 256, 28, 256, 1),
    ("patch_in 28->256 (G step)", 32, 80, 256, 28, 256, 1),
    ("ref_in   12->64", 32, 512, 90, 12, 64, 1),
    ("ref_out  64->12", 32, 512, 90, 64, 12, 1),
    ("mbd_out 384->12 g8", 512, 10, 64, 384, 12, 8),
    ("patch_out 512->12", 64, 5, 64, 512, 12, 1),
]
which = sys.argv[1] if len(sys.argv) > 1 else "fdw"
warm_s = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
dev = "cuda"
L = ops.L()
torch.manual_seed(0)


def steady(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:        # the clock ramps over tens of milliseconds
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


tot = {k: 0.0 for k in which}
floor = {k: 0.0 for k in which}
for name, N, H, W, Cc, K, G in SHAPES:
    x = torch.randn(N, H, W, Cc, device=dev)
    w = torch.randn(G * K, Cc, 1, 1, device=dev) * 0.05
    wk, wc = [], []
    for g in range(G):
        a, b = ops.weight_prep(w[g * K:(g + 1) * K].contiguous(), None, True)
        wk.append(a); wc.append(b)
    wk, wc = torch.cat(wk).detach(), torch.cat(wc).detach()
    d = ops.conv_desc(N, H, W, Cc, K, 1, 1, groups=G, precision="f32")
    y = torch.empty(N, H, W, K, device=dev)
    dy = torch.randn(N, H, W, K, device=dev)
    dx = torch.empty_like(x); dw = torch.empty_like(wk)
    nb = L.mq_conv_wgrad_workspace(C.byref(d)); ws = ops._ws_bytes(nb, x)
    fns = {"f": lambda: L.mq_conv_fprop(C.byref(d), ops.fptr(x), ops.fptr(wk), None, ops.fptr(y), ops.stream()),
           "d": lambda: L.mq_conv_dgrad(C.byref(d), ops.fptr(dy), ops.fptr(wc), ops.fptr(dx), ops.stream()),
           "w": lambda: L.mq_conv_wgrad(C.byref(d), ops.fptr(x), ops.fptr(dy), ops.fptr(dw), ops.ptr(ws), nb, ops.stream())}
    nbytes = 4.0 * N * H * W * (Cc + K)
    line = f"{name:30s} {nbytes / 1e9:5.2f} GB"
    for k in which:
        ms = steady(fns[k])
        tot[k] += ms
        floor[k] += nbytes / 6.3e9               # ms at the 6.3 TB/s a float4 copy reaches (hardware guide)
        line += f"   {k} {ms:6.3f} ms {nbytes / ms / 1e9:5.2f} TB/s"
    print(line, flush=True)
    del x, y, dy, dx, ws
print("sum: " + "   ".join(f"{k} {tot[k]:.3f} ms (at 6.3 TB/s: {floor[k]:.3f})" for k in which))
