#!/usr/bin/env python3
"""Gated dgrad (the producer layer's LeakyReLU / mask backward in the dgrad store) against the two passes it replaces,
on discriminator layer shapes: bf16-storage kernels and the f32 engine."""
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


CASES = [  # name, N, H, W, C, K, R, S, stride, pad, groups
    ("c3 D1.convs1 384->384 7x7 s2", 16, 128, 512, 384, 384, 7, 7, (2, 2), (3, 3), 1),
    ("c2 D1.convs1 32->64? 5x5 s2", 64, 80, 576, 32, 64, 5, 5, (2, 2), (2, 2), 1),
    ("c2 D1.convs2 64->128 5x5 s2", 64, 40, 288, 64, 128, 5, 5, (2, 2), (2, 2), 1),
    ("c2 D1.convs3 128->256 3x3 s1", 64, 20, 144, 128, 256, 3, 3, (1, 1), (1, 1), 1),
    ("c2 D2 32->64 3x5 s(1,2) g8", 512, 10, 576, 32, 64, 3, 5, (1, 2), (1, 2), 8),
]
for prec in ("bf16", "f32"):
    for name, N, H, W, Cc, K, R, S, st, pad, G in CASES:
        torch.manual_seed(0)
        d = ops.conv_desc(N, H, W, Cc, K, R, S, stride=st, pad=pad, groups=G, precision=prec)
        w = torch.randn(G * K, Cc, R, S, device=DEV) * 0.02
        ks, cs = zip(*[ops.weight_prep(w[g * K:(g + 1) * K].contiguous(), None, True) for g in range(G)])
        wk, wc = torch.cat(ks, 0), torch.cat(cs, 0)
        mask = torch.zeros(N, W, dtype=torch.uint8, device=DEV)
        mask[::3, W // 2:] = 1
        dmap = (H * W, W, W)
        x = torch.nn.functional.leaky_relu(torch.randn(N, H, W, Cc, device=DEV), 0.2)
        dy = torch.randn(N, d.P, d.Q, K, device=DEV)
        if prec == "bf16":
            x, dy = x.to(BF16), dy.to(BF16)
            wt = ops.h2_weight_image(d, wc, True)
            plain = lambda: ops.conv_dgrad_h2(d, dy, wt)
            gated = lambda: ops.conv_dgrad_h2(d, dy, wt, gate=(x, mask, dmap, 0.2))
        else:
            key = ops._desc_key(d)
            plain = lambda: ops._conv_backward_f32(d, key, x, wc, dy, tuple(wk.shape), True, False, False)[0]
            gated = lambda: ops._conv_backward_f32(d, key, x, wc, dy, tuple(wk.shape), True, False, False, in_gate=(mask, dmap, 0.2))[0]
        dx = plain()
        t_plain = tm(plain)
        t_pw = tm(lambda: ops.pointwise_bwd_raw(dx, x, act=ops.ACT_LEAKY, mask=mask, mask_post=True, dmap=dmap))
        t_gated = tm(gated)
        print(f"{prec:5s} {name:32s} dgrad {t_plain:7.3f} + leaky' {t_pw:7.3f} = {t_plain + t_pw:7.3f} ms   gated {t_gated:7.3f} ms"
              f"   saves {t_plain + t_pw - t_gated:+7.3f} ms", flush=True)
        del x, dy, dx, w, wk, wc
        torch.cuda.empty_cache()
