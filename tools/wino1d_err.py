#!/usr/bin/env python3
"""Error of the 1-D Winograd route (csrc/wino1d.hip) per row tile F(m, 3): the two-layer multi-bin discriminator chain of
tests/test_gpu_kernels.py::test_wino1d_discriminator_chain_matches_direct_kernels, every output and gradient against the direct
kernels AND against a float64 torch reference (relative L2 and max / rms)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mqgan_amd import ops

DEV = torch.device("cuda", 0)
H, W, G = (int(a) for a in (sys.argv[1:4] + ["10", "48", "8"][len(sys.argv) - 1:]))
torch.manual_seed(31)
N, C0, C1, C2 = 2 * G, 128, 128, 256
x0 = torch.randn(N, H, W, C0, device=DEV)
w1 = torch.randn(G * C1, C0, 3, 5, device=DEV) * 0.05
w2 = torch.randn(G * C2, C1, 3, 3, device=DEV) * 0.05
b1, b2 = torch.randn(G * C1, device=DEV) * 0.1, torch.randn(G * C2, device=DEV) * 0.1
mask1 = torch.zeros(N, W, dtype=torch.uint8, device=DEV)
mask1[1, W // 2:] = 1
mask1[-1, 5:] = 1
mask2 = ops.mask_pool(mask1, 2, ceil_mode=True)
r = None


def prep(w, K):
    ks, cs = [], []
    for gi in range(G):
        a, bb = ops.weight_prep(w[gi * K:(gi + 1) * K].contiguous(), None, True)
        ks.append(a); cs.append(bb)
    return torch.cat(ks, 0), torch.cat(cs, 0)


def run(minc, tile):
    global r
    ops.WINO1D_MIN_CHANNELS, ops.WINO1D_TILE = minc, tile
    x = x0.clone().requires_grad_(True)
    w1v, w2v = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    b1v, b2v = b1.clone().requires_grad_(True), b2.clone().requires_grad_(True)
    k1, c1 = prep(w1v, C1)
    k2, c2 = prep(w2v, C2)
    d1 = ops.conv_desc(N, H, W, C0, C1, 3, 5, pad=(1, 2), groups=G)
    d2 = ops.conv_desc(N, H, W, C1, C2, 3, 3, stride=(1, 2), pad=(1, 1), groups=G)
    dm1, dm2 = (d1.P * d1.Q, d1.Q, d1.Q), (d2.P * d2.Q, d2.Q, d2.Q)
    y1, hon = ops.conv_leaky(x, k1, c1, b1v, d1, mask1, dm1, pregate=True, chain=True)
    y2 = ops.conv_leaky(y1, k2, c2, b2v, d2, mask2, dm2, in_gate=(mask1, dm1, 0.2))
    if r is None:
        r = torch.randn(y2.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5))
    (y2 * r).sum().backward()
    return [t.detach().double().cpu() for t in (y1, y2, x.grad, w1v.grad, w2v.grad, b1v.grad, b2v.grad)]


def ref64():
    x = x0.double().cpu().requires_grad_(True)
    w1v, w2v = w1.double().cpu().requires_grad_(True), w2.double().cpu().requires_grad_(True)
    b1v, b2v = b1.double().cpu().requires_grad_(True), b2.double().cpu().requires_grad_(True)
    pg = N // G
    ys = []
    for g in range(G):
        xi = x[g * pg:(g + 1) * pg].permute(0, 3, 1, 2)
        h = F.conv2d(xi, w1v[g * C1:(g + 1) * C1], b1v[g * C1:(g + 1) * C1], padding=(1, 2))
        ys.append(h)
    h1 = torch.cat(ys, 0)
    m1 = mask1.cpu().bool()[:, None, None, :]
    y1 = F.leaky_relu(h1, 0.2).masked_fill(m1, 0.0)
    ys = []
    for g in range(G):
        ys.append(F.conv2d(y1[g * pg:(g + 1) * pg], w2v[g * C2:(g + 1) * C2], b2v[g * C2:(g + 1) * C2], stride=(1, 2), padding=(1, 1)))
    h2 = torch.cat(ys, 0)
    m2 = mask2.cpu().bool()[:, None, None, :]
    y2 = F.leaky_relu(h2, 0.2).masked_fill(m2, 0.0)
    (y2 * r.double().cpu().permute(0, 3, 1, 2)).sum().backward()
    return [y1.detach().permute(0, 2, 3, 1), y2.detach().permute(0, 2, 3, 1), x.grad, w1v.grad, w2v.grad, b1v.grad, b2v.grad]


names = ("y1", "y2", "dx", "dw1", "dw2", "db1", "db2")
direct = run(0, 0)
ref = ref64()


def show(tag, got, want):
    out = []
    for n, a, b in zip(names, got, want):
        a = a.reshape(b.shape) if a.shape != b.shape else a
        l2 = float((a - b).norm() / b.norm())
        mx = float((a - b).abs().max() / b.pow(2).mean().sqrt())
        out.append(f"{n} {l2:.1e}/{mx:.1e}")
    print(f"{tag:22s}", "  ".join(out), flush=True)


print(f"H {H} W {W} groups {G}: relative L2 / max-over-rms")
show("direct vs float64", direct, ref)
for tile in (2, 4, 5):
    got = run(128, tile)
    show(f"F({tile},3) vs float64", got, ref)
    show(f"F({tile},3) vs direct", got, direct)
    valid = ~mask1.cpu().bool()[:, None, :, None].expand_as(got[0])
    flips = ((got[0] > 0) != (direct[0] > 0)) & valid
    print(f"  LeakyReLU gate flips of layer 1 against the direct kernels: {int(flips.sum())} of {int(valid.sum())} elements;"
          f" |y1| there (direct, of its rms): {[float(v) for v in (direct[0][flips].abs() / direct[0].pow(2).mean().sqrt())][:6]}")
