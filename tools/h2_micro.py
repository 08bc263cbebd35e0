#!/usr/bin/env python3
"""HIP-event timing of the bf16-storage conv kernels (igemm_h2.hip) on the refiner / discriminator layer shapes of
BASELINE config 3 (hifimusic) and config 2 (hifispeech), next to the f32-tensor bf16 engine of igemm.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402

DEV = "cuda"
BF16 = torch.bfloat16


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


CASES = [  # name, N, H, W, C, K, R, S, stride, pad
    ("c3 mid 768->768 3x3", 16, 128, 144, 768, 768, 3, 3, (1, 1), (1, 1)),
    ("c3 ups0.c1 1152->384", 16, 256, 144, 1152, 384, 3, 3, (1, 1), (1, 1)),
    ("c3 L2 384->384", 16, 256, 144, 384, 384, 3, 3, (1, 1), (1, 1)),
    ("c3 ups1.c1 576->192", 16, 512, 144, 576, 192, 3, 3, (1, 1), (1, 1)),
    ("c3 L0 96->96", 16, 1024, 144, 96, 96, 3, 3, (1, 1), (1, 1)),
    ("c3 D1.convs1 384->384 7x7 s2", 16, 128, 512, 384, 384, 7, 7, (2, 2), (3, 3)),
    ("c2 mid 512->512", 32, 64, 90, 512, 512, 3, 3, (1, 1), (1, 1)),
    ("c2 D1.convs1 256->256 5x5 s2", 32, 80, 256, 256, 256, 5, 5, (2, 2), (2, 2)),
]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if any(a in c[0] for a in sys.argv[1:] if not a.startswith('--'))] or CASES
for name, N, H, W, Cc, K, R, S, st, pad in CASES:
    torch.manual_seed(0)
    d = ops.conv_desc(N, H, W, Cc, K, R, S, stride=st, pad=pad, precision="bf16")
    flops = 2.0 * N * d.P * d.Q * K * Cc * R * S
    x = torch.randn(N, H, W, Cc, device=DEV).to(BF16)
    w = torch.randn(K, Cc, R, S, device=DEV) * 0.02
    wk, wc = ops.weight_prep(w, None, True)
    wh, wth = ops.h2_weight_image(d, wk, False), ops.h2_weight_image(d, wc, True)
    dy = torch.randn(N, d.P, d.Q, K, device=DEV).to(BF16)
    res = {}
    res["fprop"] = tm(lambda: ops.conv_fprop_h2(d, x, wh, None))
    if "--f32out" in sys.argv:
        res["fprop(f32 out)"] = tm(lambda: ops.conv_fprop_h2(d, x, wh, None, out_dtype=torch.float32))
    res["dgrad"] = tm(lambda: ops.conv_dgrad_h2(d, dy, wth))
    res["wgrad"] = tm(lambda: ops.conv_wgrad_h2(d, x, dy, tuple(wk.shape), True))
    line = f"{name:32s} " + "  ".join(f"{k} {v:7.3f} ms {flops / v / 1e9:7.1f} TF/s" for k, v in res.items())
    if "--old" in sys.argv:
        xf, dyf = x.float(), dy.float()
        y = torch.empty(N, d.P, d.Q, K, device=DEV)
        import ctypes as C
        from mqgan_amd.ops import check, fptr, stream
        t_old = tm(lambda: check(ops.L().mq_conv_fprop(C.byref(d), fptr(xf), fptr(wk), None, fptr(y), stream()), "f"))
        line += f"   | old fprop {t_old:7.3f} ms {flops / t_old / 1e9:7.1f} TF/s"
        del xf, dyf, y
    print(line, flush=True)
    del x, dy, w, wk, wc, wh, wth
    torch.cuda.empty_cache()
