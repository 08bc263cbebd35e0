#!/usr/bin/env python3
"""Random shapes through the bf16-storage kernels (mq_conv_fprop_h2 / mq_conv_dgrad_h2 / mq_conv_wgrad_h2) against a float64
convolution of the same bf16-representable operands: output (f32 and bf16 store), input gradient, weight and bias gradient, as
relative L2 and as the largest element error over the tensor's rms (a wrong edge tile or tap is an error of order 1 in a few
elements: the L2 norm of a whole tensor can absorb it).  Channel counts are drawn so that every kernel family gets its share: 64-
and 32-channel chunks, the shared-row kernels (3 / 5 / 7 stride-1 column taps), the 96-channel kernel, strided layers (input
gradient class by class through the shared-row kernel), batch groups, dilation, ragged tiles.  `fuzz_h2.py [cases] [seed]`"""
import collections, os, random, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.ops import check, fptr, stream  # noqa: E402

DEV, BF16 = "cuda", torch.bfloat16
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
last = lambda: ops.L().mq_last_kernel().decode().split("(")[0]


def one(N, H, W, C, K, R, S, stride, pad, dil, G):
    torch.manual_seed(rnd.randrange(1 << 30))
    rb = lambda v: v.bfloat16().double()
    x = rb(torch.randn(N, H, W, C, dtype=torch.float64)).requires_grad_(True)
    w = rb(torch.randn(G * K, C, R, S, dtype=torch.float64) * 0.2).requires_grad_(True)
    b = torch.randn(G * K, dtype=torch.float64).float().double().requires_grad_(True)
    d = ops.conv_desc(N, H, W, C, K, R, S, stride=stride, pad=pad, dil=dil, groups=G, precision="bf16")
    if not ops.h2_supported(d):
        return None
    ng = N // G
    ys = [F.conv2d(x[g * ng:(g + 1) * ng].permute(0, 3, 1, 2), w[g * K:(g + 1) * K], b[g * K:(g + 1) * K], stride=stride, padding=pad,
                   dilation=dil) for g in range(G)]
    y_ref = torch.cat(ys, 0).permute(0, 2, 3, 1)
    r = torch.randn_like(y_ref).bfloat16().double()
    (y_ref * r).sum().backward()
    wd = w.detach().float().to(DEV)
    ks, cs = zip(*(ops.weight_prep(wd[g * K:(g + 1) * K].contiguous(), None, True) for g in range(G)))
    wk, wc = torch.cat(ks, 0), torch.cat(cs, 0)
    xh, rh = x.detach().to(DEV).to(BF16), r.to(DEV).to(BF16)
    res, names = {}, []
    res["y"] = (ops.conv_fprop_h2(d, xh, ops.h2_weight_image(d, wk, False), b.detach().float().to(DEV), out_dtype=torch.float32), y_ref.detach())
    names.append(last())
    res["dx"] = (ops.conv_dgrad_h2(d, rh, ops.h2_weight_image(d, wc, True), out_dtype=torch.float32), x.grad)
    names.append(last())
    if ops.h2_wgrad_supported(d):
        dw, db = ops.conv_wgrad_h2(d, xh, rh, tuple(wk.shape), True)
        names.append(last())
        dws = []
        for g in range(G):
            v = torch.zeros(K, C, R * S, device=DEV)
            dv = torch.empty_like(v)
            check(ops.L().mq_weight_prep_bwd(0, fptr(dw[g * K:(g + 1) * K].contiguous()), fptr(v), None, None, K, C, R * S, fptr(dv), None,
                                             stream()), "mq_weight_prep_bwd")
            dws.append(dv.view(K, C, R, S))
        res["dw"] = (torch.cat(dws, 0), w.grad)
        res["db"] = (db, b.grad)
    out = {}
    for k, (got, want) in res.items():
        got, want = got.double().cpu(), want.double()
        rms = float(want.pow(2).mean().sqrt()) or 1.0
        out[k] = (float((got - want).norm() / (want.norm() or 1.0)), float((got - want).abs().max()) / rms)
    return out, names


CH = [32, 64, 96, 128, 160, 192, 256, 288, 384]
worst = collections.defaultdict(float)
seen = collections.Counter()
done = 0
while done < cases:
    kind = rnd.choice(["s1", "s1", "96", "strided", "strided", "cols", "dil", "1x1"])
    G = rnd.choice([1, 1, 1, 2])
    dil = (1, 1)
    if kind == "s1":
        C, K, R, S, stride = rnd.choice(CH), rnd.choice(CH), rnd.choice([1, 3]), 3, (1, 1)
    elif kind == "96":
        C, K, R, S, stride = rnd.choice([96, 288]), rnd.choice([96, 192, 288]), rnd.choice([1, 3]), 3, (1, 1)
    elif kind == "strided":
        C, K = rnd.choice([64, 128, 192, 256]), rnd.choice([64, 128, 192])
        R, S = rnd.choice([(3, 3), (5, 5), (7, 7), (3, 5)])
        stride = rnd.choice([(2, 2), (1, 2), (2, 1)])
    elif kind == "cols":
        C, K, R, S, stride = rnd.choice([64, 128, 256]), rnd.choice([64, 128, 192]), rnd.choice([1, 3]), rnd.choice([5, 7]), (1, 1)
    elif kind == "dil":
        C, K, R, S, stride = rnd.choice([64, 96]), rnd.choice([32, 64]), 3, 3, (1, 1)
        dil = (rnd.choice([1, 2]), rnd.choice([1, 2, 3]))
    else:
        C, K, R, S, stride = rnd.choice(CH), rnd.choice(CH), 1, 1, (1, 1)
    pad = (dil[0] * (R // 2), dil[1] * (S // 2))
    N = G * rnd.randint(1, 3)
    H = 1 if R == 1 and rnd.random() < 0.5 else rnd.randint(2, 14)
    W = rnd.randint(16, 150)
    got = one(N, H, W, C, K, R, S, stride, pad, dil, G)
    if got is None:
        continue
    errs, names = got
    done += 1
    for n in names:
        seen[n] += 1
    bad = [k for k, (l2, mx) in errs.items() if l2 > (4e-5 if k != "db" else 1e-4) or mx > 4e-4]
    for k, (l2, mx) in errs.items():
        worst[k] = max(worst[k], l2)
        worst[k + " max"] = max(worst[k + " max"], mx)
    print(f"{done:3d} {kind:7s} N{N} {H}x{W} C{C}->K{K} k{R}x{S} s{stride} d{dil} g{G}: "
          + " ".join(f"{k} {l2:.1e}/{mx:.1e}" for k, (l2, mx) in errs.items()) + "  [" + ", ".join(names) + "]"
          + ("   <-- ABOVE THE BOUND: " + ",".join(bad) if bad else ""), flush=True)
    if bad:
        sys.exit(1)
print("kernels exercised:", dict(seen))
print("worst (relative L2 / largest element over rms):", {k: f"{v:.2e}" for k, v in worst.items()})
print(f"FUZZ OK {cases} cases")
