#!/usr/bin/env python3
"""Fused ConvBlock2D at BASELINE config 2 (B 32 x T 512 x C0 512), train mode (dropout 0.1) and eval mode: forward and backward
HIP-event times, and the (pixel, hidden unit) rate against the chip's vector-instruction issue rate."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.preencoder import ConvBlock2D  # noqa: E402

DEV = "cuda"
B, T, C0 = 32, 512, 512
torch.manual_seed(0)
m = ConvBlock2D(1, C0, kernel_size=5, depthwise=True, act="aptx").to(DEV)
x = torch.randn(B, T, C0, device=DEV, requires_grad=True)
mask = torch.zeros(B, T, dtype=torch.uint8, device=DEV)
dy = torch.randn(B, T, C0, device=DEV)


def tm(fn, it=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


pairs = B * T * C0 * C0
for mode in ("train", "eval"):
    m.train(mode == "train")
    with torch.no_grad():
        f = tm(lambda: m(x, mask))

    def fb():
        x.grad = None
        m(x, mask).backward(dy)
    fbt = tm(fb)
    print(f"{mode}: forward {f:.3f} ms ({pairs / f / 1e9:.2f} T pairs/s), backward {fbt - f:.3f} ms ({pairs / (fbt - f) / 1e9:.2f} T pairs/s)")
