#!/usr/bin/env python3
"""Does it pay to run a Winograd layer in image chunks whose transformed tensors (V, M) stay in the 256 MiB Infinity Cache?
Forward of one refiner layer of config 2 (input transform -> 42-group GEMM -> output transform), whole batch against chunks of
n images; HIP-event time per layer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402

DEV = "cuda"


def tm(fn, it=5):
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
    ("L0 64->64", 32, 512, 90, 64, 64),
    ("ups2.c1 192->64", 32, 512, 90, 192, 64),
    ("L1 128->128", 32, 256, 90, 128, 128),
    ("L2 256->256", 32, 128, 90, 256, 256),
    ("L3 512->512", 32, 64, 90, 512, 512),
]
for name, N, H, W, Cc, K in CASES:
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Cc, device=DEV)
    w = torch.randn(K, Cc, 3, 3, device=DEV) * 0.02
    wk, wc = ops.weight_prep(w, None, True)
    wk, wc = wk.detach(), wc.detach()
    bias = torch.randn(K, device=DEV)
    dy = torch.randn(N, H, W, K, device=DEV)
    line = f"{name:18s}"
    for n in (32, 16, 8, 4, 2, 1):
        d = ops.conv_desc(n, H, W, Cc, K, 3, 3, pad=(1, 1), precision="f32")
        key = ops._desc_key(d)
        vm_mib = 4 * 42 * n * (H // 4) * (W // 5) * (Cc + K) / 2**20

        xs = [x[i:i + n].clone().requires_grad_(True) for i in range(0, N, n)]

        def fwd():
            with torch.no_grad():
                for xi in xs:
                    ops.conv(xi, wk, wc, bias, d)

        def fwd_dgrad():
            for i, xi in enumerate(xs):
                xi.grad = None
                ops.conv(xi, wk, wc, bias, d).backward(dy[i * n:(i + 1) * n])
        tf = tm(fwd)
        line += f" | n={n:2d} V+M {vm_mib:6.0f} MiB fwd {tf:6.3f} dgrad {tm(fwd_dgrad) - tf:6.3f}"
    print(line, flush=True)
    del x, w, wk, wc, dy
    torch.cuda.empty_cache()
