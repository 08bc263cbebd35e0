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
    bias = torch.randn(K, device=DEV)
    dy = torch.randn(N, H, W, K, device=DEV)
    line = f"{name:18s}"
    for n in (32, 16, 8, 4, 2, 1):
        d = ops.conv_desc(n, H, W, Cc, K, 3, 3, pad=(1, 1), precision="f32")
        key = ops._desc_key(d)
        vm_mib = 4 * 42 * n * (H // 4) * (W // 5) * (Cc + K) / 2**20

        def fwd():
            for i in range(0, N, n):
                ops._wino_conv(d, x[i:i + n], wk, bias, Cc, K, False, key)

        def dgrad():
            for i in range(0, N, n):
                Z, Vd, part = ops._wino_dy_fused(dy[i:i + n], d)
                ops._wino_conv(d, None, wk, None, K, Cc, True, key, V=Vd)
        line += f" | n={n:2d} V+M {vm_mib:6.0f} MiB fwd {tm(fwd):6.3f} dgrad {tm(dgrad):6.3f}"
    print(line, flush=True)
    del x, w, wk, wc, dy
    torch.cuda.empty_cache()
