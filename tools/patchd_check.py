#!/usr/bin/env python3
"""The patch discriminator alone (config-2 widths, small ragged batch), forward + backward with the trainer's pregate chain:
parameter gradients and the input gradient with the polyphase / column Winograd forms on against the direct kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.discriminators import MelSpectrogramPatchDiscriminator2D, run_discriminators  # noqa: E402

DEV = "cuda"
torch.manual_seed(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = MelSpectrogramPatchDiscriminator2D(80, hidden_channels=[256, 256, 384, 512, 512], kernel_sizes=[5, 5, 5, 3, 3, 3],
                                       stride=[[1, 2], [2, 2], [2, 2], [2, 1], [2, 1], [2, 1]]).to(DEV)
d.train() if os.environ.get('D_TRAIN') == '1' else d.eval()
kind = sys.argv[2] if len(sys.argv) > 2 else "randn"
if len(sys.argv) > 3:
    ops.WINO_TILE = sys.argv[3]
if kind == "mel":
    import numpy as np
    from mqgan_amd.train import synthetic_mel
    x0 = torch.from_numpy(np.stack([synthetic_mel(T, 80, 700 + b) for b in range(2)])).to(DEV)
else:
    x0 = torch.randn(2, T, 80, device=DEV)
print("input", kind, "mean %.3f std %.3f" % (float(x0.mean()), float(x0.std())), "tile", ops.WINO_TILE)
lens = torch.tensor([T, (T * 41) // 64], dtype=torch.int32, device=DEV)
r = None
res = {}
_orig_ok = ops._wino_s2d_ok
ONLYK = int(os.environ.get("ONLYK", "0"))
if ONLYK:
    ops._wino_s2d_ok = lambda dd: _orig_ok(dd) and dd.K == ONLYK
    print("polyphase form only for the layer with K =", ONLYK)
for name, s2d, ww in (("direct", 0, 0), ("s2d", 128, 0)):
    ops.WINO_S2D_MIN_CHANNELS, ops.WINOW_MIN_CHANNELS = s2d, ww
    x = x0.clone().requires_grad_(True)
    for p in d.parameters():
        p.grad = None
    logits, mask, feats = run_discriminators([d], x, lens, True, pregate=True)
    if r is None:
        r = torch.randn_like(logits) if os.environ.get("R_CONST") != "1" else torch.full_like(logits, -0.0125)
        rf = [torch.randn_like(f[0]) for f in feats]
    loss = (logits * r).sum()
    loss.backward()
    res[name] = {"x": x.grad.clone(), "logits": logits.detach().clone()}
    for n_, p in d.named_parameters():
        if p.grad is not None:
            res[name][n_] = p.grad.clone()
    for i, f in enumerate(feats):
        res[name][f"feat{i}"] = f[0].detach().clone()
for name in ("s2d",):
    print("==", name, "against direct")
    for k, b in res["direct"].items():
        a = res[name][k].double().flatten()
        b = b.double().flatten()
        e = a - b
        print(f"  {k:28s} rel {float(e.norm() / (b.norm() + 1e-300)):.2e} cos {float(e @ b / (e.norm() * b.norm() + 1e-300)):+.3f} "
              f"max/max {float(e.abs().max() / b.abs().max()):.2e} nonzero-diff {int((e != 0).sum())}/{e.numel()}")
if os.environ.get("ERRMAP") == "1":
    a, b = res["s2d"]["x"].double(), res["direct"]["x"].double()          # (B, T, 80)
    e = (a - b).abs()
    scale = float(b.abs().max())
    torch.set_printoptions(linewidth=250, precision=1, sci_mode=True)
    print("error of the input gradient / max|ref|, maximum over mel bins, per (item, frame):")
    print(e.amax(dim=2) / scale)
    print("maximum over frames, per (item, mel bin):")
    print(e.amax(dim=1) / scale)
    print("reference magnitude per frame (max over bins):")
    print(b.abs().amax(dim=2) / scale)
