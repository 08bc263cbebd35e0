#!/usr/bin/env python3
"""Bandwidth of the fused mask + activation + dropout kernels on a refiner-sized tensor.
usage: pointwise_micro.py [act(0 none,1 aptx,2 leaky)] [p_drop]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mqgan_amd import ops

act = int(sys.argv[1]) if len(sys.argv) > 1 else 1
p = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
B, T, Fq, Cc = 32, 512, 90, 64
x = torch.randn(B, T, Fq, Cc, device="cuda", requires_grad=True)
mask = torch.zeros(B, T, dtype=torch.uint8, device="cuda")
beta = torch.ones(1, device="cuda"); gamma = torch.full((1,), 0.5, device="cuda")
ops.manual_seed(0)
kw = dict(act=act, mask=mask, mask_post=True, dmap=(T * Fq, Fq, 1), p_drop=p, training=True,
          beta=beta if act == 1 else None, gamma=gamma if act == 1 else None)
y = ops.pointwise(x, **kw)
g = torch.randn_like(y)
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
nbytes = x.numel() * 4
fwd = timeit(lambda: ops.pointwise(x.detach(), **kw))
def bwd():
    xx = x.detach().requires_grad_(True)
    yy = ops.pointwise(xx, **kw)
    yy.backward(g)
both = timeit(bwd)
print(f"act={act} p={p}: fwd {fwd:.3f} ms = {2 * nbytes / fwd / 1e6:.0f} GB/s (8 B/elem);  fwd+bwd {both:.3f} ms -> bwd {both - fwd:.3f} ms = "
      f"{3 * nbytes / (both - fwd) / 1e6:.0f} GB/s (12 B/elem)")
