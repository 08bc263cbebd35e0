#!/usr/bin/env python3
"""Error of the Winograd routes per tile shape against float64 (CPU): one 3x3 / stride-1 layer (plan form winograd2d) and one 5x5 /
stride-2 layer in the polyphase form (plan form polyphase-winograd2d, LeakyReLU branch fixed to the tested path's own) - output, input
gradient, weight gradient as relative L2 and max over rms.  `wino_err.py [N H W C K]`."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mqgan_amd import ops

DEV = torch.device("cuda", 0)
N, H, W, Cc, K = (int(a) for a in (sys.argv[1:6] + ["2", "64", "90", "256", "256"][len(sys.argv) - 1:]))


def errs(got, want):
    out = []
    for n, a, b in zip(("y", "dx", "dw"), got, want):
        a = a.double().cpu().reshape(b.shape)
        out.append(f"{n} {float((a - b).norm() / b.norm()):.1e}/{float((a - b).abs().max() / b.pow(2).mean().sqrt()):.1e}")
    return "  ".join(out)


torch.manual_seed(0)
x0 = torch.randn(N, H, W, Cc, device=DEV)
w0 = torch.randn(K, Cc, 3, 3, device=DEV) * 0.05
r0 = torch.randn(N, H, W, K, device=DEV)
xd = x0.double().cpu().requires_grad_(True)
wd = w0.double().cpu().requires_grad_(True)
yd = F.conv2d(xd.permute(0, 3, 1, 2), wd, padding=1).permute(0, 2, 3, 1)
(yd * r0.double().cpu()).sum().backward()
ref = [yd.detach(), xd.grad, wd.grad]
print(f"3x3 stride 1, N {N} {H}x{W} C {Cc} -> K {K}: relative L2 / max-over-rms against float64")
for tile in (0, 3, 4, 45, 6):
    ops.WINO_MIN_CHANNELS, ops.WINO_TILE = (0, "auto") if tile == 0 else (128, tile)
    x = x0.clone().requires_grad_(True)
    w = w0.clone().requires_grad_(True)
    wk, wc = ops.weight_prep(w, None, True)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, pad=(1, 1))
    y = ops.conv(x, wk, wc, None, d)
    (y * r0).sum().backward()
    print(f"  {'direct' if tile == 0 else 'tile %s' % tile:10s}", errs([y.detach(), x.grad, w.grad], ref), flush=True)

# 5x5 stride 2 (no activation effects: slope 1.0 would need another kernel path, so compare with the branch fixed)
H2, W2 = 2 * (H // 2), 2 * (W // 2)
x0 = torch.randn(N, H2, W2, Cc, device=DEV)
w0 = torch.randn(K, Cc, 5, 5, device=DEV) * 0.03
b0 = torch.zeros(K, device=DEV)
r0 = torch.randn(N, H2 // 2, W2 // 2, K, device=DEV)
print(f"5x5 stride 2 + LeakyReLU, N {N} {H2}x{W2} C {Cc} -> K {K}")
for tile in (0, 3, 4, 45, 6):
    ops.WINO_S2D_MIN_CHANNELS, ops.WINO_TILE = (0, "auto") if tile == 0 else (128, tile)
    x = x0.clone().requires_grad_(True)
    w = w0.clone().requires_grad_(True)
    wk, wc = ops.weight_prep(w, None, True)
    d = ops.conv_desc(N, H2, W2, Cc, K, 5, 5, stride=(2, 2), pad=(2, 2))
    y = ops.conv_leaky(x, wk, wc, b0, d, None, (d.P * d.Q, d.Q, d.Q))
    y = y[0] if isinstance(y, tuple) else y
    (y * r0).sum().backward()
    xd = x0.double().cpu().requires_grad_(True)
    wd = w0.double().cpu().requires_grad_(True)
    h = F.conv2d(xd.permute(0, 3, 1, 2), wd, stride=2, padding=2).permute(0, 2, 3, 1)
    yd = torch.where((y > 0).cpu(), h, 0.2 * h)
    (yd * r0.double().cpu()).sum().backward()
    print(f"  {'direct' if tile == 0 else 'tile %s' % tile:10s}", errs([y.detach(), x.grad, w.grad], [yd.detach(), xd.grad, wd.grad]), flush=True)
