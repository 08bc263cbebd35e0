#!/usr/bin/env python3
"""What Winograd would cost in accuracy on the bf16-storage engine: a 3x3 / stride-1 layer with
bf16-representable x and w (what the direct bf16 kernels see: exact products, f32 accumulation, bf16 output) against
F(2x2, 3x3) and F(4x4, 3x3) with f32 transforms and V = B^T d B, U = G g G^T ROUNDED TO bf16 for the bf16 MFMA (M in f32 or
rounded to bf16 as well), all against float64.  CPU, numpy/torch; prints relative L2 and max/rms of the output error."""
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def wino_mats(m):
    if m == 2:
        BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1.]])
        G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
        AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1.]])
    else:   # F(4, 3) on the points of csrc/wino_tables.h (0, 1, -1, 2, -1/2, inf)
        BT = torch.tensor([[1, 1.5, -2, -1.5, 1, 0], [0, 1 / 3, 5 / 6, 1 / 6, -1 / 3, 0], [0, 1 / 3, 1 / 6, -5 / 6, 1 / 3, 0],
                           [0, -1 / 30, -1 / 15, 1 / 30, 1 / 15, 0], [0, -32 / 15, 16 / 15, 32 / 15, -16 / 15, 0],
                           [0, 1, 1.5, -2, -1.5, 1.]])
        G = torch.tensor([[1, 0, 0], [1, 1, 1], [1, -1, 1], [1, 2, 4], [1, -.5, .25], [0, 0, 1.]])
        AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -.5, 0], [0, 1, 1, 4, .25, 0], [0, 1, -1, 8, -.125, 1.]])
    return BT, G, AT


def wino_conv(x, w, m, round_m):
    """x (C, H, W) f32 (bf16-representable), w (K, C, 3, 3); H, W multiples of m; 'same' padding."""
    BT, G, AT = wino_mats(m)
    n = m + 2
    C, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(1, n, m).unfold(2, n, m)                     # (C, th, tw, n, n)
    V = bf(torch.einsum("ia,cxyab,jb->cxyij", BT, tiles, BT))        # f32 transform, bf16 storage
    U = bf(torch.einsum("ia,kcab,jb->kcij", G, w, G))
    M = torch.einsum("cxyij,kcij->kxyij", V.double(), U.double()).float()      # exact products, (ideal) f32 accumulation
    if round_m:
        M = bf(M)
    Y = torch.einsum("ai,kxyij,bj->kxyab", AT, M, AT)
    K, th, tw = Y.shape[:3]
    return Y.permute(0, 1, 3, 2, 4).reshape(K, th * m, tw * m)


C, K, H, W = 256, 64, 16, 16
x = bf(torch.randn(C, H, W))
w = bf(torch.randn(K, C, 3, 3) * 0.05)
ref = F.conv2d(x.double()[None], w.double(), padding=1)[0]
rms = float(ref.pow(2).mean().sqrt())


def report(name, y):
    e = y.double() - ref
    print(f"{name:58s} rel L2 {float(e.norm() / ref.norm()):.2e}   max/rms {float(e.abs().max()) / rms:.2e}")


direct = F.conv2d(x.double()[None], w.double(), padding=1)[0].float()           # exact products, f32 result
report("direct, bf16 operands, f32 output", direct)
report("direct, bf16 operands, bf16 output (what the H2 kernels store)", bf(direct))
for m in (2, 4):
    for rm in (False, True):
        y = wino_conv(x, w, m, rm)
        report(f"F({m}x{m},3x3), V and U bf16, M {'bf16' if rm else 'f32'}, f32 output", y)
        report(f"F({m}x{m},3x3), V and U bf16, M {'bf16' if rm else 'f32'}, bf16 output", bf(y))
