#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Runs only in the build container (needs /root/reference, read-only).  Nothing
from the reference is copied: this script imports its modules, feeds them seeded
inputs and stores inputs / weights / outputs / gradients as numeric arrays
(.npz) plus a key->shape manifest (.json).  The fixtures travel to the GPU box,
the reference does not.

einx and wandb are absent from this image; they are replaced by in-memory
stand-ins before import (einx.where is only reached in FSQ's train-mode noise
branch, quantizer.py:151,160; wandb only by train.py's logging calls).

Usage:  python tests/golden/gen_golden.py            (rewrites every fixture)
"""
import json
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MQGAN_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    einx = types.ModuleType("einx")
    einx.where = lambda pat, m, a, b: torch.where(
        m.reshape(m.shape[0], *([1] * (b.dim() - 1))), a, b)
    sys.modules["einx"] = einx
    wandb = types.ModuleType("wandb")
    for name in ("init", "watch", "log", "finish"):
        setattr(wandb, name, lambda *a, **k: None)
    wandb.Image = lambda *a, **k: None
    sys.modules["wandb"] = wandb
    sys.path.insert(0, REF)


def sd_np(module, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def perturb(module, seed, scale=0.3):
    """Move weight-norm gains / biases / APTx params off their init so that
    every parameter has a non-trivial effect and gradient."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith("original0") or name.endswith("weight_g"):
                p.mul_(1.0 + scale * torch.randn(p.shape, generator=g))
            elif name.endswith("bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith(".beta") or name.endswith(".gamma"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))


def grads_np(module, prefix):
    out = {}
    for k, p in module.named_parameters():
        out[prefix + k] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
    return out


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"  {name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ----------------------------------------------------------------------------- G1
def gen_fsq():
    from quantizer import FSQ
    torch.manual_seed(7)
    levels = [8, 5, 5, 5]
    q = FSQ(levels=levels).eval()
    z = torch.randn(4096, 4) * 2.0
    # edge cases: zeros, huge magnitudes, tiny values
    z[0] = 0.0
    z[1] = torch.tensor([50.0, -50.0, 1e-8, -1e-8])
    z[2] = torch.tensor([-9.0, 9.0, 0.25, -0.25])
    z = z.reshape(8, 512, 4).clone().requires_grad_(True)
    codes, idx = q(z)
    wgt = torch.randn(8, 512, 4)
    (codes * wgt).sum().backward()
    with torch.no_grad():
        b = q.bound(z)
        margin = ((b - b.floor()) - 0.5).abs()
        half_width = (q._levels // 2).float()
        bounded_norm = b / half_width
        all_idx = torch.arange(q.codebook_size)
        dec = q.indices_to_codes(all_idx.reshape(1, -1))
    save("g1_fsq.npz", levels=np.array(levels, np.int32), z=z.detach().numpy(), codes=codes.detach().numpy(),
         idx=idx.numpy(), wgt=wgt.numpy(), grad_z=z.grad.numpy(), bound=b.numpy(), margin=margin.numpy(),
         bounded_norm=bounded_norm.numpy(), implicit_codebook=q.implicit_codebook.numpy(),
         decode_all=dec.numpy()[0])
    # a larger index-only set to measure CPU<->GPU flip rate on the box (int16 to stay small)
    torch.manual_seed(8)
    zb = torch.randn(1, 16384, 4) * 2.0
    with torch.no_grad():
        _, ib = q(zb)
    save("g1b_fsq_idx16k.npz", seed=np.array([8]), z=zb.numpy()[0], idx=ib.numpy()[0].astype(np.int16))


# ----------------------------------------------------------------------------- G2
def gen_convblock2d():
    from preencoder import ConvBlock2D
    torch.manual_seed(21)
    m = ConvBlock2D(1, 16, kernel_size=5, depthwise=True, act="aptx")
    perturb(m, 22)
    m.eval()
    x = torch.randn(2, 16, 24, requires_grad=True)
    lens = torch.tensor([24, 17])
    mask = (torch.arange(24)[None, :] >= lens[:, None]).unsqueeze(1)  # (B,1,T) True = padded
    out, hidden = m(x, mask, return_hidden=True)
    r = torch.randn_like(out)
    (out * r).sum().backward()
    save("g2_convblock2d.npz", x=x.detach().numpy(), lens=lens.numpy(), out=out.detach().numpy(),
         hidden=hidden.detach().numpy(), r=r.numpy(), grad_x=x.grad.numpy(),
         **sd_np(m, "sd/"), **grads_np(m, "grad/"))


# ----------------------------------------------------------------------------- G3
def gen_resblocks():
    from attentions import ResidualBlock1D
    arrays = {}
    specs = [("a", 16, 24, 5, False), ("b", 24, 24, 3, False), ("c", 24, 16, 7, True), ("d", 16, 16, 3, True)]
    for tag, cin, cout, k, causal in specs:
        torch.manual_seed(30 + ord(tag))
        m = ResidualBlock1D(cin, cout, kernel_size=k, dropout=0.1, act="taptx", causal=causal, norm="weight")
        perturb(m, 31 + ord(tag))
        m.eval()
        T = 37
        x = torch.randn(2, cin, T, requires_grad=True)
        lens = torch.tensor([T, 20])
        mask = (torch.arange(T)[None, :] >= lens[:, None]).unsqueeze(1)
        out = m(x, x_mask=mask)
        r = torch.randn_like(out)
        (out * r).sum().backward()
        p = f"{tag}/"
        arrays.update({p + "cfg": np.array([cin, cout, k, int(causal)]), p + "x": x.detach().numpy(),
                       p + "lens": lens.numpy(), p + "out": out.detach().numpy(), p + "r": r.numpy(),
                       p + "grad_x": x.grad.numpy()})
        arrays.update(sd_np(m, p + "sd/"))
        arrays.update(grads_np(m, p + "grad/"))
    save("g3_resblock1d.npz", **arrays)


# ----------------------------------------------------------------------------- G4
def gen_refiner():
    from preencoder import UNetRefiner
    torch.manual_seed(41)
    m = UNetRefiner(1, base_ch=8, depth=3, dropout=0.1, input_out_channels=[18, 16])
    perturb(m, 42)
    m.eval()
    T = 21
    x = torch.randn(2, 1, T, 18, requires_grad=True)
    lens = torch.tensor([T, 13])
    mask = torch.arange(T)[None, :] >= lens[:, None]  # (B,T)
    out = m(x, mask)
    r = torch.randn_like(out)
    (out * r).sum().backward()
    save("g4_refiner.npz", x=x.detach().numpy(), lens=lens.numpy(), out=out.detach().numpy(), r=r.numpy(),
         grad_x=x.grad.numpy(), **sd_np(m, "sd/"), **grads_np(m, "grad/"))


# ----------------------------------------------------------------------------- G5
PATCH_CFG = dict(hidden_channels=[8, 8, 12, 16, 16], kernel_sizes=[5, 5, 5, 3, 3, 3],
                 stride=[[1, 2], [2, 2], [2, 2], [2, 1], [2, 1], [2, 1]])
MBD_CFG = dict(hidden_channels=[8, 8, 16, 16, 24], kernel_sizes=[7, 5, 3, 3, 3, 3], n_bins=8, n_no_strides=2)


def _flat(prefix, obj, arrays):
    """store nested lists/tuples of tensors under indexed names"""
    if isinstance(obj, torch.Tensor):
        arrays[prefix] = obj.detach().numpy().copy()
    else:
        arrays[prefix + "/len"] = np.array([len(obj)])
        for i, o in enumerate(obj):
            _flat(f"{prefix}/{i}", o, arrays)


def gen_discriminators():
    from discriminators import MelSpectrogramPatchDiscriminator2D, MultiBinDiscriminator
    arrays = {}
    mel, T = 16, 40
    lens = torch.tensor([T, 29])
    for tag, ctor in (("patch", lambda: MelSpectrogramPatchDiscriminator2D(mel, **PATCH_CFG)),
                      ("mbd", lambda: MultiBinDiscriminator(mel, **MBD_CFG))):
        torch.manual_seed(50 + len(tag))
        d = ctor()
        # N(0, 0.02) init gives ~1e-2 singular values; keep as the reference initialises it
        with torch.no_grad():
            for n, p in d.named_parameters():
                if n.endswith("bias"):
                    p.add_(0.05 * torch.randn_like(p))
        p = tag + "/"
        arrays.update(sd_np(d, p + "sd0/"))
        # pass 1: train mode (one power iteration per conv), with gradients
        d.train()
        x1 = torch.randn(2, T, mel, requires_grad=True)
        logits, pmask, feats = d(x1, lens, return_features=True)
        arrays[p + "x1"] = x1.detach().numpy()
        _flat(p + "t/logits", logits, arrays)
        _flat(p + "t/mask", pmask, arrays)
        _flat(p + "t/feats", feats, arrays)
        if isinstance(logits, torch.Tensor):
            logits_l = [logits]
        else:
            logits_l = logits
        gen = torch.Generator().manual_seed(99)
        rs = [torch.randn(l.shape, generator=gen) for l in logits_l]
        sum((l * r).sum() for l, r in zip(logits_l, rs)).backward()
        _flat(p + "t/r", rs, arrays)
        arrays[p + "t/grad_x"] = x1.grad.numpy().copy()
        arrays.update(grads_np(d, p + "t/grad/"))
        arrays.update(sd_np(d, p + "sd1/"))  # u, v after the power iteration
        d.zero_grad()
        # pass 2: eval mode (no power iteration), gradients wrt input and weights
        d.eval()
        x2 = torch.randn(2, T, mel, requires_grad=True)
        logits, pmask, feats = d(x2, lens, return_features=True)
        arrays[p + "x2"] = x2.detach().numpy()
        _flat(p + "e/logits", logits, arrays)
        _flat(p + "e/mask", pmask, arrays)
        _flat(p + "e/feats", feats, arrays)
        logits_l = [logits] if isinstance(logits, torch.Tensor) else logits
        sum((l * r).sum() for l, r in zip(logits_l, rs)).backward()
        arrays[p + "e/grad_x"] = x2.grad.numpy().copy()
        arrays.update(grads_np(d, p + "e/grad/"))
        arrays.update(sd_np(d, p + "sd2/"))  # must equal sd1 (eval leaves u, v alone)
    arrays["lens"] = lens.numpy()
    save("g5_discriminators.npz", **arrays)


# ----------------------------------------------------------------------------- G6
def gen_losses():
    from losses import LSGANLoss, MaskedMelLoss
    import train as ref_train
    arrays = {}
    torch.manual_seed(61)
    gl = LSGANLoss()
    n_calls = 9
    for i in range(n_calls):
        real = torch.randn(2, 1, 3, 8, requires_grad=True)
        fake = (torch.randn(2, 1, 3, 8) * 0.7 + 0.2).requires_grad_(True)
        mask = torch.rand(2, 1, 3, 8) > 0.3
        if i == 4:
            mask[:] = False  # empty-mask branch of _masked_mse
        loss = gl.discriminator_loss(real, fake, mask, mask)
        if loss.requires_grad:
            loss.backward()
        p = f"d{i}/"
        arrays.update({p + "real": real.detach().numpy(), p + "fake": fake.detach().numpy(), p + "mask": mask.numpy(),
                       p + "loss": loss.detach().numpy(),
                       p + "grad_real": (real.grad if real.grad is not None else torch.zeros_like(real)).numpy(),
                       p + "grad_fake": (fake.grad if fake.grad is not None else torch.zeros_like(fake)).numpy(),
                       p + "ema_real": gl.ema_real.numpy().copy(), p + "ema_fake": gl.ema_fake.numpy().copy()})
    for i in range(3):
        fake = torch.randn(2, 1, 3, 8, requires_grad=True)
        mask = torch.rand(2, 1, 3, 8) > 0.3
        loss = gl.generator_loss(fake, mask)
        loss.backward()
        p = f"g{i}/"
        arrays.update({p + "fake": fake.detach().numpy(), p + "mask": mask.numpy(), p + "loss": loss.detach().numpy(),
                       p + "grad_fake": fake.grad.numpy()})
    # masked mel loss, group 1 and 16 (train.py:217-218)
    x = torch.randn(3, 40, 32, requires_grad=True)
    y = torch.randn(3, 40, 32)
    lens = torch.tensor([40, 29, 7])
    for g in (1, 16):
        x.grad = None
        loss = MaskedMelLoss("mse", group_size=g)(x, y, lens)
        loss.backward()
        arrays[f"mel{g}/loss"] = loss.detach().numpy()
        arrays[f"mel{g}/grad_x"] = x.grad.numpy().copy()
    x.grad = None
    loss = MaskedMelLoss("charbonnier", group_size=1)(x, y, lens)
    loss.backward()
    arrays["melc/loss"] = loss.detach().numpy()
    arrays["melc/grad_x"] = x.grad.numpy().copy()
    arrays.update({"mel/x": x.detach().numpy(), "mel/y": y.numpy(), "mel/lens": lens.numpy()})
    # masked_mae (feature matching, train.py:38-45)
    pf = torch.randn(2, 6, 5, 9, requires_grad=True)
    tf = torch.randn(2, 6, 5, 9)
    mk = (torch.rand(2, 1, 5, 9) > 0.6)
    l = ref_train.masked_mae(pf, tf, mk)
    l.backward()
    arrays.update({"mae/pred": pf.detach().numpy(), "mae/target": tf.numpy(), "mae/mask": mk.numpy(),
                   "mae/loss": l.detach().numpy(), "mae/grad_pred": pf.grad.numpy()})
    save("g6_losses.npz", **arrays)


# ----------------------------------------------------------------------------- G7 / G9
TINY = dict(mel_channels=16, channels=[16, 16, 24], kernel_sizes=[3, 5], fsq_levels=[8, 5, 5, 5], dropout=0.1,
            refiner_base_channels=8, refiner_depth=3, refiner_hidden_proj_divisor=8)


def tiny_config(use_fm):
    return {
        "model": {"mel_channels": 16},
        "training": {"lr": 1e-4, "beta1": 0.9, "beta2": 0.999, "lr_d_factor": 1.15, "d_beta1": 0.5, "d_beta2": 0.999,
                     "warmup_steps": 1000, "discriminator_train_start_epoch": 0,
                     "loss_weights": {"fm_lambda": 0.25, "Gloss_lambda": 15.0, "recon_lambda": 15.0},
                     "use_fm_loss": use_fm, "seed": 42, "no_cuda": True, "pretrained": None},
    }


def summarize(module, prefix):
    out = {}
    for k, v in module.state_dict().items():
        v = v.detach().double()
        out[prefix + k] = np.array([v.sum().item(), v.norm().item()])
    return out


def gen_trainstep(use_fm, fname):
    """Two optimisation steps driven by the reference's own Trainer methods
    (train.py:380-501), generator in eval() so that dropout / FSQ noise are off."""
    import train as ref_train
    from preencoder import PreEncoder
    from discriminators import MelSpectrogramPatchDiscriminator2D, MultiBinDiscriminator
    from losses import LSGANLoss, MaskedMelLoss
    torch.manual_seed(42)
    cfg = tiny_config(use_fm)
    t = ref_train.Trainer.__new__(ref_train.Trainer)
    t.config = cfg
    t.device = torch.device("cpu")
    t.generator = PreEncoder(**TINY)
    with torch.no_grad():
        t.generator.q_in_proj.weight.mul_(40.0)  # spread the FSQ indices (random init collapses to one code)
    perturb(t.generator, 43, scale=0.2)
    t.patch_discriminator = MelSpectrogramPatchDiscriminator2D(16, **PATCH_CFG)
    t.multibin_discriminator = MultiBinDiscriminator(16, **MBD_CFG)
    t.optimizer_g, t.optimizer_d = t._init_optimizers()
    t.scheduler_g = t._init_schedulers()
    t.gan_loss = LSGANLoss()
    t.recon_loss_all = MaskedMelLoss("mse")
    t.recon_loss_group = MaskedMelLoss("mse", group_size=16)
    t.scaler_g = ref_train.GradScaler(enabled=False)
    t.scaler_d = ref_train.GradScaler(enabled=False)
    t.epoch = 1
    arrays = {}
    # record the PRE-clip total norms that clip_grad_norm_ computes (train.py:403,487 discard them)
    norms = []
    orig_clip = torch.nn.utils.clip_grad_norm_

    def recording_clip(params, max_norm, *a, **k):
        n = orig_clip(params, max_norm, *a, **k)
        norms.append(float(n))
        return n
    torch.nn.utils.clip_grad_norm_ = recording_clip
    arrays.update(sd_np(t.generator, "g0/"))
    arrays.update(sd_np(t.patch_discriminator, "p0/"))
    arrays.update(sd_np(t.multibin_discriminator, "m0/"))
    t.generator.eval()
    t.patch_discriminator.train()
    t.multibin_discriminator.train()
    B, T = 2, 40
    rng = np.random.default_rng(1234)
    lens = torch.tensor([T, 29], dtype=torch.int32)
    for step in range(2):
        mel = np.clip(rng.standard_normal((B, T, 16)) * 2.0 - 4.0, -11.5129, 5.0).astype(np.float32)
        mel[1, 29:] = 0.0
        x = torch.from_numpy(mel)
        recon_pre, recon_post = t.generator(x, lens)
        loss_d = t._train_discriminator(x, recon_post, lens)
        g_losses = t._train_generator(x, recon_pre, recon_post, lens)
        p = f"s{step}/"
        arrays[p + "x"] = mel
        arrays[p + "recon_pre"] = recon_pre.detach().numpy()
        arrays[p + "recon_post"] = recon_post.detach().numpy()
        arrays[p + "loss_d"] = np.array(loss_d)
        for k, v in g_losses.items():
            arrays[p + k] = np.array(v)
        arrays[p + "preclip_norm_d"] = np.array(norms[-2])
        arrays[p + "preclip_norm_g"] = np.array(norms[-1])
        arrays[p + "lr"] = np.array(t.scheduler_g.get_last_lr()[0])
        arrays[p + "ema"] = np.array([t.gan_loss.ema_real.item(), t.gan_loss.ema_fake.item()])
        # post-clip gradient norms (clip_grad_norm_ scaled them in place)
        arrays[p + "gnorm_g"] = np.array(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in t.generator.parameters()
                                                       if q.grad is not None)).item())
        dpar = list(t.patch_discriminator.parameters()) + list(t.multibin_discriminator.parameters())
        arrays[p + "gnorm_d"] = np.array(torch.sqrt(sum((q.grad.double() ** 2).sum() for q in dpar
                                                       if q.grad is not None)).item())
        arrays.update(summarize(t.generator, p + "gsum/"))
        arrays.update(summarize(t.patch_discriminator, p + "psum/"))
        arrays.update(summarize(t.multibin_discriminator, p + "msum/"))
    torch.nn.utils.clip_grad_norm_ = orig_clip
    arrays["lens"] = lens.numpy()
    # a few full tensors after two steps
    for k in ("refiner.post.parametrizations.weight.original1", "encoder_blocks.1.conv2.parametrizations.weight.original0",
              "decoder_blocks.0.conv1.weight_v", "out_proj.weight", "pre.pw.parametrizations.weight.original0"):
        arrays["gfinal/" + k] = t.generator.state_dict()[k].numpy().copy()
    arrays["pfinal/convs.1.weight_orig"] = t.patch_discriminator.state_dict()["convs.1.weight_orig"].numpy().copy()
    arrays["pfinal/convs.1.weight_u"] = t.patch_discriminator.state_dict()["convs.1.weight_u"].numpy().copy()
    save(fname, **arrays)
    return t


def gen_encdec(trainer):
    g = trainer.generator.eval()
    torch.manual_seed(91)
    B, T = 2, 40
    x = torch.randn(B, T, 16) * 2.0 - 4.0
    lens = torch.tensor([T, 31])
    mask = (torch.arange(T)[None, :] >= lens[:, None]).unsqueeze(1)
    with torch.no_grad():
        idx = g.encode(x, mask)
        dec, hid = g.decode(idx, mask, return_hidden=True)
        rec, post = g(x, lens)
        idx_nomask = g.encode(x, None)
        dec_nomask = g.decode(idx_nomask, None)
    save("g9_encdec.npz", x=x.numpy(), lens=lens.numpy(), idx=idx.numpy(), dec=dec.numpy(), hid=hid.numpy(),
         fwd_recon=rec.numpy(), fwd_post=post.numpy(), idx_nomask=idx_nomask.numpy(), dec_nomask=dec_nomask.numpy(),
         **sd_np(g, "sd/"))


# ----------------------------------------------------------------------------- G8
def gen_manifest():
    import yaml
    from preencoder import PreEncoder
    from discriminators import MelSpectrogramPatchDiscriminator2D, MultiBinDiscriminator
    man = {}
    for cfg_name, mels in (("hifispeech", (80, 128)), ("hifimusic", (128, 160))):
        with open(os.path.join(REF, "configs", f"model_config_{cfg_name}.yaml")) as f:
            cfg = yaml.safe_load(f)
        gen_cfg = cfg["model"]["generator"]
        for mel in mels:
            g = PreEncoder(mel_channels=mel, channels=gen_cfg["channels"], kernel_sizes=gen_cfg["kernel_sizes"],
                           dropout=gen_cfg["dropout"], fsq_levels=gen_cfg["fsq_levels"],
                           refiner_base_channels=gen_cfg.get("refiner_base_channels", 128),
                           refiner_depth=gen_cfg.get("refiner_depth", 3))
            pc = cfg["model"]["discriminator_patch"]
            d1 = MelSpectrogramPatchDiscriminator2D(mel, hidden_channels=pc["hidden_channels"],
                                                    kernel_sizes=pc["kernel_sizes"], stride=pc["strides"])
            mc = cfg["model"]["discriminator_multibin"]
            d2 = MultiBinDiscriminator(mel, hidden_channels=mc["hidden_channels"], kernel_sizes=mc["kernel_sizes"],
                                       n_bins=mc["n_bins"], n_no_strides=mc["n_no_strides"])
            entry = {}
            for tag, m in (("generator", g), ("patch", d1), ("multibin", d2)):
                entry[tag] = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()}
                entry[tag + "_params"] = sum(p.numel() for p in m.parameters())
            man[f"{cfg_name}@{mel}"] = entry
    man["checkpoint_keys"] = ["epoch", "model_state_dict", "optimizer_g_state_dict", "optimizer_d_state_dict",
                              "scaler_g_state_dict", "scaler_d_state_dict", "config"]
    path = os.path.join(OUT, "g8_manifest.json")
    with open(path, "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print(f"  g8_manifest.json: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    _install_stubs()
    torch.set_num_threads(1)
    print("generating golden fixtures from", REF)
    gen_fsq()
    gen_convblock2d()
    gen_resblocks()
    gen_refiner()
    gen_discriminators()
    gen_losses()
    t = gen_trainstep(False, "g7_trainstep.npz")
    gen_encdec(t)
    gen_trainstep(True, "g7b_trainstep_fm.npz")
    gen_manifest()


if __name__ == "__main__":
    main()
