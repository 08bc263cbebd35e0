#!/usr/bin/env python3
"""The weight gradients of the hifimusic refiner's 96-channel-chunk layers (BASELINE config 3, full size) on
conv_wgrad_h2s96_kernel against the gathered-tile kernels they ran on until r6 (MQGAN_H2_NO_WGRAD96=1, read per call), same
process, alternating, each after 0.3 s of the same launch (clock warm-up).  Prints ms, TF/s and the largest difference."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402

DEV, BF16 = "cuda", torch.bfloat16


def tm(fn, it=8, warm_s=0.3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        fn()
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it


CASES = [("L0 96->96", 16, 1024, 144, 96, 96), ("L0 ups 288->96", 16, 1024, 144, 288, 96), ("L1 down 96->192", 16, 512, 144, 96, 192),
         ("L1 192->192", 16, 512, 144, 192, 192), ("L1 ups 576->192", 16, 512, 144, 576, 192), ("L2 down 192->384", 16, 256, 144, 192, 384)]
tot = {"new": 0.0, "old": 0.0}
for name, N, H, W, Cc, K in CASES:
    torch.manual_seed(0)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, pad=(1, 1), precision="bf16")
    flops = 2.0 * N * H * W * K * Cc * 9
    x = torch.randn(N, H, W, Cc, device=DEV).to(BF16)
    dy = torch.randn(N, H, W, K, device=DEV).to(BF16)
    shape = (K, 9 * Cc)
    out = {}
    for rep in range(2):
        for mode in ("new", "old"):
            if mode == "old":
                os.environ["MQGAN_H2_NO_WGRAD96"] = "1"
            else:
                os.environ.pop("MQGAN_H2_NO_WGRAD96", None)
            dw, db = ops.conv_wgrad_h2(d, x, dy, shape, True)
            kern = ops.L().mq_last_kernel().decode()
            ms = tm(lambda: ops.conv_wgrad_h2(d, x, dy, shape, True))
            out[mode] = (ms, kern, dw, db)
            if rep == 1:
                tot[mode] += ms
        if rep == 1:
            (a, ka, dwa, dba), (b, kb, dwb, dbb) = out["new"], out["old"]
            rel = float((dwa - dwb).abs().max()) / float(dwb.pow(2).mean().sqrt())
            relb = float((dba - dbb).abs().max()) / float(dbb.pow(2).mean().sqrt())
            print(f"{name:18s} {ka:32s} {a:6.3f} ms {flops / a / 1e9:7.1f} TF/s | {kb:40s} {b:6.3f} ms {flops / b / 1e9:7.1f} TF/s | "
                  f"max |dw diff| / rms {rel:.2e}, bias {relb:.2e}", flush=True)
os.environ.pop("MQGAN_H2_NO_WGRAD96", None)
print(f"sum over the six layers (wgrad + slab / bias reduce): new {tot['new']:.3f} ms, old {tot['old']:.3f} ms")
