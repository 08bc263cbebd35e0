#!/usr/bin/env python3
"""The Winograd form mq_conv_plan picks (F(4x5, 3x3) for these images) against the direct f32 kernels on the refiner's 3x3 layers of BASELINE config 2 (hifispeech):
forward + input gradient + weight gradient of one layer, HIP-event time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402

DEV = "cuda"


def tm(fn, it=3):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


CASES = [  # name, N, H(T), W(F), C, K
    ("c2 L3 mid 512->512", 32, 72, 80, 512, 512),
    ("c2 L2 256->256", 32, 144, 80, 256, 256),
    ("c2 ups0.c1 768->256", 32, 144, 80, 768, 256),
    ("c2 L1 128->128", 32, 288, 80, 128, 128),
    ("c2 ups1.c1 384->128", 32, 288, 80, 384, 128),
    ("c2 L0 64->64", 32, 576, 80, 64, 64),
    ("c2 ups2.c1 192->64", 32, 576, 80, 192, 64),
]
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
for name, N, H, W, Cc, K in CASES:
    torch.manual_seed(0)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, pad=(1, 1), precision=prec)
    key = ops._desc_key(d)
    flops = 2.0 * N * H * W * K * Cc * 9
    x = torch.randn(N, H, W, Cc, device=DEV)
    w = torch.randn(K, Cc, 3, 3, device=DEV) * 0.02
    wk, wc = ops.weight_prep(w, None, True)
    dy = torch.randn(N, H, W, K, device=DEV)
    bias = torch.randn(K, device=DEV)
    res = {}

    def direct():
        y = ops.ConvFn.apply(x, wk, wc, bias, key)
        return y
    res["direct fprop"] = tm(direct)
    res["direct bwd"] = tm(lambda: ops._conv_backward_f32(d, key, x, wc, dy, tuple(wk.shape), True, True, True))
    res["wino fprop"] = tm(lambda: ops.conv(x, wk, wc, bias, d))

    xg, wkg, bg = x.clone().requires_grad_(True), wk.detach().clone().requires_grad_(True), bias.clone().requires_grad_(True)

    def wino_fb():
        xg.grad = wkg.grad = bg.grad = None
        ops.conv(xg, wkg, wc, bg, d).backward(dy)
    res["wino bwd"] = tm(wino_fb) - res["wino fprop"]
    dsum, wsum = res["direct fprop"] + res["direct bwd"], res["wino fprop"] + res["wino bwd"]
    print(f"{name:24s} direct {res['direct fprop']:7.3f} + {res['direct bwd']:7.3f} = {dsum:7.3f} ms ({3 * flops / dsum / 1e9:6.1f} TF/s)   "
          f"winograd {res['wino fprop']:7.3f} + {res['wino bwd']:7.3f} = {wsum:7.3f} ms   x{dsum / wsum:4.2f}", flush=True)
    del x, dy, w, wk, wc
    torch.cuda.empty_cache()
