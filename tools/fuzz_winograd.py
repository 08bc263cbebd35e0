#!/usr/bin/env python3
"""Random shapes through the Winograd routes of ops.conv (2-D tiles 3x3 / 4x4 / 4x5 and the per-image choice; the column forms
F(4,3) / F(4,5) / F(2,7) with random padding, row taps and strides, batch groups) against the direct kernels: output, input
gradient, weight gradient, bias gradient as relative L2 and as the largest element error over the tensor's rms (a wrong edge
tile is an error of order 1 in a few elements: the L2 norm of a whole tensor can absorb it).  `fuzz_winograd.py [cases] [seed]`"""
import os, sys, random
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from mqgan_amd import ops  # noqa: E402

DEV = "cuda"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def run(desc_args, G, C, K, R, S, N, H, W, route):
    torch.manual_seed(rnd.randrange(1 << 30))
    x0 = torch.randn(N, H, W, C, device=DEV)
    w0 = torch.randn(G * K, C, R, S, device=DEV) * 0.1
    b0 = torch.randn(G * K, device=DEV)
    outs = {}
    for leg in ("direct", route):
        on = leg != "direct"
        ops.WINO_MIN_CHANNELS = 128 if (on and route.startswith("2d")) else 0
        ops.WINOW_MIN_CHANNELS = 128 if (on and route == "col") else 0
        if on and route.startswith("2d"):
            ops.WINO_TILE = route[2:] or "auto"
        x = x0.clone().requires_grad_(True)
        w = w0.clone().requires_grad_(True)
        b = b0.clone().requires_grad_(True)
        ks, cs = [], []
        for gi in range(G):
            a, bb = ops.weight_prep(w[gi * K:(gi + 1) * K].contiguous(), None, True)
            ks.append(a); cs.append(bb)
        d = ops.conv_desc(N, H, W, C, K, R, S, groups=G, precision="f32", **desc_args)
        y = ops.conv(x, torch.cat(ks, 0), torch.cat(cs, 0), b, d)
        name = y.grad_fn.__class__.__name__
        if leg == "direct":
            r = torch.randn_like(y)
        (y * r).sum().backward()
        outs[leg] = (name, [y.detach(), x.grad, w.grad, b.grad])
    errs = [float((a - b_).norm() / b_.norm()) for a, b_ in zip(outs[route][1], outs["direct"][1])]
    maxs = [float((a - b_).abs().max() / b_.pow(2).mean().sqrt()) for a, b_ in zip(outs[route][1], outs["direct"][1])]
    return outs[route][0], errs, maxs


worst = worst_max = 0.0
for i in range(cases):
    if i % 2 == 0:
        route = "2d" + rnd.choice(["", "3", "4", "45", "6"])
        C, K = rnd.choice([128, 144, 256]), rnd.choice([128, 160, 256])
        N, H, W = rnd.randint(1, 3), rnd.randint(1, 23), rnd.randint(1, 27)
        args, G, R, S = dict(pad=(1, 1)), 1, 3, 3
    else:
        route = "col"
        S = rnd.choice([3, 5, 7])
        R = rnd.choice([1, 1, 3])
        G = rnd.choice([1, 1, 2])
        C, K = rnd.choice([128, 144]), rnd.choice([128, 160])
        N, H, W = G * rnd.randint(1, 2), (1 if R == 1 else rnd.randint(2, 9)), rnd.randint(2, 45)
        sh = rnd.choice([1, 2]) if R == 3 else 1
        if rnd.random() < 0.4:
            args = dict(stride=(sh, 1), pad=(R // 2, S - 1), out_hw=((H + 2 * (R // 2) - R) // sh + 1, W))        # causal
        else:
            args = dict(stride=(sh, 1), pad=(R // 2, S // 2))
    name, errs, maxs = run(args, G, C, K, R, S, N, H, W, route)
    worst, worst_max = max(worst, max(errs)), max(worst_max, max(maxs))
    flag = "" if max(errs) < 4e-5 and max(maxs) < 2e-4 else "   <-- above 4e-5 (L2) / 2e-4 (element)"
    print(f"{i:3d} {route:5s} {name:22s} N{N} {H}x{W} C{C}->K{K} k{R}x{S} g{G} {args}: " + " ".join(f"{e:.1e}" for e in errs)
          + " | max/rms " + " ".join(f"{e:.1e}" for e in maxs) + flag, flush=True)
print(f"worst against the direct kernels over {cases} cases: relative L2 {worst:.2e}, largest element error {worst_max:.2e} of the rms")
