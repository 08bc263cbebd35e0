#!/usr/bin/env python3
"""Per-parameter gradients of one config-2-architecture optimisation step (small ragged batch) on the HIP path against
oracle.train_step.OracleTrainer, for the switches given on the command line (NAME=VALUE pairs set on mqgan_amd.ops, e.g.
WINO_MIN_CHANNELS=0 WINOW_MIN_CHANNELS=0 WINO1D_MIN_CHANNELS=0 WINO_S2D_MIN_CHANNELS=0 WINO_TILE=3).  Diagnostic twin of
tests/test_gpu_trainstep.py::test_hifispeech_architecture_step_matches_oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.train import Trainer  # noqa: E402
import test_gpu_trainstep as T  # noqa: E402

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    old = getattr(ops, k)
    setattr(ops, k, type(old)(v) if not isinstance(old, str) else v)
    print("ops.%s = %r" % (k, getattr(ops, k)))
o = T._oracle_hifispeech_inputs()
cfg, x, lens = o["cfg"], o["x"], o["lens"]
tr = Trainer(cfg, device="cuda", build_data=False)
tr.generator.load_state_dict(o["Pg"], strict=True)
tr.patch_discriminator.load_state_dict(o["Pp"], strict=True)
tr.multibin_discriminator.load_state_dict(o["Pm"], strict=True)
tr.arena_g.rebind(); tr.arena_d.rebind()
tr.generator.eval(); tr.patch_discriminator.train(); tr.multibin_discriminator.train()
tr.epoch = 1
from mqgan_amd import discriminators as D_
SHARE = os.environ.get("SHARE_GATES", "1") == "1"      # the oracle takes the LeakyReLU branches of the HIP run
if SHARE:
    D_.GATE_CAPTURE = []
out = tr.train_step(x.cuda(), lens.cuda())
o = T._oracle_hifispeech_step(D_.GATE_CAPTURE if SHARE else None)
D_.GATE_CAPTURE = None
otr, want = o["otr"], o["want"]
print("LeakyReLU branches shared with the oracle:", SHARE)
for k in ("loss_d", "loss_g_total", "loss_recon_pre", "loss_recon_post", "loss_gan"):
    print(f"{k:16s} hip {float(out[k]):.8f} oracle {float(want[k]):.8f} rel {abs(float(out[k]) - float(want[k])) / abs(float(want[k])):.1e}")
print("norm_d", float(tr.optimizer_d.grad_norm()), otr.last_norm_d, " norm_g", float(tr.optimizer_g.grad_norm()), otr.last_norm_g)
got_g = T._arena_grads(tr.arena_g, tr.generator)
got_d = T._arena_grads(tr.arena_d, tr.patch_discriminator, tr.multibin_discriminator)
for tag, got, ref in (("G", got_g, otr.last_grads_g), ("D", got_d, otr.last_grads_d)):
    for k, gw in ref.items():
        gg, gw = got[k].double(), gw.double()
        scale = float(gw.abs().max())
        if scale == 0.0:
            continue
        rel = float((gg - gw).norm() / gw.norm())
        mx = float((gg - gw).abs().max()) / scale
        # projection of the error on the reference: a pure scale error shows as cos ~ +-1
        e = (gg - gw).flatten()
        cos = float((e @ gw.flatten()) / (e.norm() * gw.norm() + 1e-300))
        print(f"{tag} {k:70s} rel {rel:.2e} max {mx:.2e} cos(err, ref) {cos:+.3f} n {gw.numel()}")
import statistics
for tag, got, ref, pref in (("G refiner", got_g, otr.last_grads_g, "refiner."), ("G decoder", got_g, otr.last_grads_g, "decoder_blocks."),
                            ("G encoder", got_g, otr.last_grads_g, "encoder_blocks."), ("D patch", got_d, otr.last_grads_d, "0:"),
                            ("D multibin", got_d, otr.last_grads_d, "1:")):
    rels = []
    for k, gw in ref.items():
        if not k.startswith(pref) or float(gw.abs().max()) == 0.0 or gw.numel() < 64:
            continue
        rels.append(float((got[k].double() - gw.double()).norm() / gw.double().norm()))
    print(f"SUMMARY {tag:12s} tensors {len(rels):3d} median rel {statistics.median(rels):.2e} max {max(rels):.2e} above 1e-4: {sum(r > 1e-4 for r in rels)}")
for tag, got, ref, hip_norm, o_norm in (("G", got_g, otr.last_grads_g, float(tr.optimizer_g.grad_norm()), otr.last_norm_g),
                                        ("D", got_d, otr.last_grads_d, float(tr.optimizer_d.grad_norm()), otr.last_norm_d)):
    extra = {k: float(v.double().norm()) for k, v in got.items() if k not in ref}
    print(tag, "HIP-only tensors with a gradient:", {k: v for k, v in extra.items() if v > 0})
    n_ref = sum(float(v.double().pow(2).sum()) for v in ref.values()) ** 0.5
    n_got = sum(float(v.double().pow(2).sum()) for v in got.values()) ** 0.5
    print(f"{tag} norm: HIP kernel {hip_norm:.7f}  float64 over the HIP arena {n_got:.7f} | oracle clip_grad_norm_ {o_norm:.7f}  float64 over the oracle's grads {n_ref:.7f}")
