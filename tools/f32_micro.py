#!/usr/bin/env python3
"""HIP-event timing of the f32 conv kernels on the layer shapes of BASELINE config 2 (hifispeech).  Run twice, without and
with MQGAN_D32=1, to compare the register-staged kernels (default) with the LDS-DMA ones (igemm_h2.hip: conv_d32_*)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.ops import check, fptr, ptr, stream  # noqa: E402

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


CASES = [
    ("c2 mid 512->512", 32, 64, 90, 512, 512, 3, 3, (1, 1), (1, 1)),
    ("c2 ups0.c1 768->256", 32, 128, 90, 768, 256, 3, 3, (1, 1), (1, 1)),
    ("c2 L1 128->128", 32, 256, 90, 128, 128, 3, 3, (1, 1), (1, 1)),
    ("c2 L0 64->64", 32, 512, 90, 64, 64, 3, 3, (1, 1), (1, 1)),
    ("c2 D1.convs1 256->256 5x5 s2", 32, 80, 256, 256, 256, 5, 5, (2, 2), (2, 2)),
]
print("MQGAN_D32 =", os.environ.get("MQGAN_D32"))
for name, N, H, W, Cc, K, R, S, st, pad in CASES:
    torch.manual_seed(0)
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride=st, pad=pad, precision="f32")
    flops = 2.0 * N * d.P * d.Q * K * Cc * R * S
    x = torch.randn(N, H, W, Cc, device=DEV)
    w = torch.randn(K, Cc, R, S, device=DEV) * 0.02
    wk, wc = ops.weight_prep(w, None, True)
    dy = torch.randn(N, d.P, d.Q, K, device=DEV)
    y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(wk)
    db = torch.empty(K, device=DEV)
    nb = ops.L().mq_conv_wgrad_bias_workspace(C.byref(d))
    ws = ops._ws_bytes(nb, x)
    t1 = tm(lambda: check(ops.L().mq_conv_fprop(C.byref(d), fptr(x), fptr(wk), None, fptr(y), stream()), "f"))
    t2 = tm(lambda: check(ops.L().mq_conv_dgrad(C.byref(d), fptr(dy), fptr(wc), fptr(dx), stream()), "d"))
    t3 = tm(lambda: check(ops.L().mq_conv_wgrad_bias(C.byref(d), fptr(x), fptr(dy), fptr(dw), fptr(db), ptr(ws), nb, stream()), "w"))
    print(f"{name:32s} fprop {t1:7.3f} ms {flops / t1 / 1e9:6.1f} TF/s  dgrad {t2:7.3f} ms {flops / t2 / 1e9:6.1f} TF/s  "
          f"wgrad {t3:7.3f} ms {flops / t3 / 1e9:6.1f} TF/s", flush=True)
