#!/usr/bin/env python3
"""Single-shape conv micro-benchmark (fprop / dgrad / wgrad) for profiling one kernel in isolation.
usage: conv_micro.py <shape-name> [iters] [fdw] [f32|f32x3|bf16|bf16h]   (bf16h: bf16 operand copies, cast outside the timing)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mqgan_amd import ops

SHAPES = {
    "mid":   dict(N=32, H=64, W=90, C=512, K=512, R=3, S=3, stride=(1, 1), pad=(1, 1), groups=1),
    "up2":   dict(N=32, H=512, W=90, C=192, K=64, R=3, S=3, stride=(1, 1), pad=(1, 1), groups=1),
    "d1c1":  dict(N=32, H=80, W=256, C=256, K=256, R=5, S=5, stride=(2, 2), pad=(2, 2), groups=1),
    "d2c1":  dict(N=256, H=10, W=512, C=128, K=128, R=3, S=5, stride=(1, 1), pad=(1, 2), groups=8),
    "midsq": dict(N=32, H=64, W=64, C=512, K=512, R=3, S=3, stride=(1, 1), pad=(1, 1), groups=1),     # 4096 tiles = 4 full rounds
    "mid1":  dict(N=32, H=64, W=64, C=4608, K=512, R=1, S=1, stride=(1, 1), pad=(0, 0), groups=1),    # same GEMM, no taps
    "mid8":  dict(N=64, H=64, W=64, C=512, K=512, R=3, S=3, stride=(1, 1), pad=(1, 1), groups=1),     # 8 full rounds
    "c3r1":  dict(N=16, H=1024, W=144, C=96, K=96, R=3, S=3, stride=(1, 1), pad=(1, 1), groups=1),     # hifimusic refiner levels
    "c3r2":  dict(N=16, H=512, W=144, C=192, K=192, R=3, S=3, stride=(1, 1), pad=(1, 1), groups=1),
    "c3mid": dict(N=16, H=128, W=144, C=768, K=768, R=3, S=3, stride=(1, 1), pad=(1, 1), groups=1),
    "c3d2":  dict(N=256, H=16, W=1024, C=128, K=128, R=3, S=5, stride=(1, 1), pad=(1, 2), groups=8),
    "enc":   dict(N=32, H=1, W=512, C=512, K=512, R=1, S=3, stride=(1, 1), pad=(0, 1), groups=1),
}
name = sys.argv[1] if len(sys.argv) > 1 else "mid"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
which = sys.argv[3] if len(sys.argv) > 3 else "fdw"
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
c = SHAPES[name]
dev = "cuda"
torch.manual_seed(0)
G = c["groups"]
x = torch.randn(c["N"], c["H"], c["W"], c["C"], device=dev, requires_grad=True)
w = (torch.randn(G * c["K"], c["C"], c["R"], c["S"], device=dev) * 0.05).requires_grad_(True)
wk, wc = [], []
for g in range(G):
    a, b = ops.weight_prep(w[g * c["K"]:(g + 1) * c["K"]].contiguous(), None, True)
    wk.append(a); wc.append(b)
wk, wc = torch.cat(wk), torch.cat(wc)
d = ops.conv_desc(c["N"], c["H"], c["W"], c["C"], c["K"], c["R"], c["S"], stride=c["stride"], pad=c["pad"], groups=G,
                  precision="bf16" if prec == "bf16h" else prec)
y = ops.conv(x, wk, wc, None, d)
dy = torch.randn_like(y)
flops = ops._conv_flops(d)
import ctypes as C
L = ops.L()
dx = torch.empty_like(x); dw = torch.empty_like(wk)
nb = L.mq_conv_wgrad_workspace(C.byref(d)); ws = ops._ws_bytes(nb, x)
fns = {"f": lambda: L.mq_conv_fprop(C.byref(d), ops.fptr(x), ops.fptr(wk), None, ops.fptr(y), ops.stream()),
       "d": lambda: L.mq_conv_dgrad(C.byref(d), ops.fptr(dy), ops.fptr(wc), ops.fptr(dx), ops.stream()),
       "w": lambda: L.mq_conv_wgrad(C.byref(d), ops.fptr(x), ops.fptr(dy), ops.fptr(dw), ops.ptr(ws), nb, ops.stream())}
if prec == "bf16h":
    BF = torch.bfloat16
    x_h, wk_h, dy_h, wc_h = ops.cast_bf16(x.detach()), ops.cast_bf16(wk.detach()), ops.cast_bf16(dy), ops.cast_bf16(wc.detach())
    fns["f"] = lambda: L.mq_conv_fprop_h(C.byref(d), ops.ptr(x_h, BF), ops.ptr(wk_h, BF), None, ops.fptr(y), ops.stream())
    fns["d"] = lambda: L.mq_conv_dgrad_h(C.byref(d), ops.ptr(dy_h, BF), ops.ptr(wc_h, BF), ops.fptr(dx), ops.stream())
    fns["c"] = lambda: ops.cast_bf16(x.detach())
for k in which:
    fn = fns[k]
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"{name} {prec} {k}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
