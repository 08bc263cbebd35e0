"""CPU restatement of the optimisation step of train.py (rows T1-T3 of SURVEY.md
section 8a) on top of ``oracle.nn``.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows train.py:380-410 (_train_discriminator), :412-501 (_train_generator),
:312-329 (optimisers, LambdaLR warm-up on G only), :523-530 (one generator
forward shared by both steps).  Adam / clip_grad_norm_ / LambdaLR are torch's
own, as in the reference.
"""
from typing import Dict

import torch

from . import nn as O


class OracleTrainer:
    """State + step.  ``Pg``/``Pp``/``Pm`` are parameter dicts (reference key
    names) for the generator, the patch discriminator and the multi-bin
    discriminator; ``cfg`` is the reference's YAML schema (``model`` and
    ``training`` sections are read)."""

    def __init__(self, Pg, Pp, Pm, cfg, gen_training=False):
        self.Pg, self.Pp, self.Pm, self.cfg = Pg, Pp, Pm, cfg
        m, t = cfg["model"], cfg["training"]
        self.levels = m["generator"]["fsq_levels"]
        self.p_drop = m["generator"].get("dropout", 0.1)
        self.gen_training = gen_training
        pc, mc = m["discriminator_patch"], m["discriminator_multibin"]
        self.patch_strides = O.patch_strides(pc["strides"], len(pc["kernel_sizes"]))
        self.n_bins, self.n_no_strides = mc["n_bins"], mc["n_no_strides"]
        self.g_params = [p for p in Pg.values() if p.requires_grad]
        self.d_params = [p for p in list(Pp.values()) + list(Pm.values()) if p.requires_grad]
        self.opt_g = torch.optim.Adam(self.g_params, lr=t["lr"], betas=(t["beta1"], t["beta2"]))
        self.opt_d = torch.optim.Adam(self.d_params, lr=t["lr"] * t["lr_d_factor"], betas=(t["d_beta1"], t["d_beta2"]))
        warm = t["warmup_steps"]
        self.sched_g = torch.optim.lr_scheduler.LambdaLR(self.opt_g, lambda s: min((s + 1) / warm, 1.0))
        self.gan = O.LSGANState()
        self.d_training = True          # train.py:505-506 sets train() per epoch, :415-416 eval() in the G step
        self.last_norm_d = self.last_norm_g = None
        # gradients as backward() left them (BEFORE clip_grad_norm_ scales them in place), by parameter name: what the parity
        # tests compare the HIP path's gradient arenas with
        self.last_grads_d: Dict[str, torch.Tensor] = {}
        self.last_grads_g: Dict[str, torch.Tensor] = {}

    @staticmethod
    def _snapshot(*dicts):
        return {("%d:" % i if len(dicts) > 1 else "") + k: p.grad.detach().clone()
                for i, P in enumerate(dicts) for k, p in P.items() if p.requires_grad and p.grad is not None}

    # -- discriminator calls -------------------------------------------------
    def _gates(self):
        """Parity tests: ``self.gate_source`` = per discriminator call of a step (in this class's call order) the list over
        layers of LeakyReLU branches (B or n_bins * B, C, H, W) another implementation took; None = the sign of the pre-activation."""
        src = getattr(self, "gate_source", None)
        if src is None:
            return None
        g = src[self._gate_call]
        self._gate_call += 1
        return g

    def _d1(self, x, lens, feats=False):
        return O.patch_discriminator(self.Pp, "", x, lens, self.patch_strides, self.d_training, feats, self._gates())

    def _d2(self, x, lens, feats=False):
        return O.multibin_discriminator(self.Pm, "", x, lens, self.n_bins, self.n_no_strides, self.d_training, feats, self._gates())

    # -- train.py:380-410 ------------------------------------------------------
    def train_discriminator(self, real, recon, lens):
        self.opt_d.zero_grad()
        rl, rm = self._d1(real, lens)[:2]
        fl, fm = self._d1(recon.detach(), lens)
        loss_d1 = O.lsgan_d_loss(self.gan, rl, fl, rm, fm)
        rl2, rm2 = self._d2(real, lens)[:2]
        fl2, fm2 = self._d2(recon.detach(), lens)
        loss_mbd = torch.tensor(0.0)
        for i in range(len(rl2)):
            loss_mbd = loss_mbd + O.lsgan_d_loss(self.gan, rl2[i], fl2[i], rm2[0], fm2[0])   # QUIRK: mask of bin 0
        loss_mbd = loss_mbd / len(rl2)
        loss_d = loss_d1 + loss_mbd
        loss_d.backward()
        self.last_grads_d = self._snapshot(self.Pp, self.Pm)        # keys "0:<patch name>" / "1:<multi-bin name>"
        clip = self.cfg["training"].get("clip_grad_norm", 1.0)
        if clip:
            self.last_norm_d = float(torch.nn.utils.clip_grad_norm_(self.d_params, clip))
        self.opt_d.step()
        return loss_d.item()

    # -- train.py:412-501 ------------------------------------------------------
    def train_generator(self, real, recon_pre, recon_post, lens, d_active=True):
        t = self.cfg["training"]
        self.opt_g.zero_grad()
        self.d_training = False
        lp = O.masked_mel_loss(recon_pre, real, lens, 1) + 0.25 * O.masked_mel_loss(recon_pre, real, lens, 16)
        lq = O.masked_mel_loss(recon_post, real, lens, 1) + 0.25 * O.masked_mel_loss(recon_post, real, lens, 16)
        loss_gan = torch.tensor(0.0)
        loss_fm = torch.tensor(0.0)
        lam_g = lam_fm = 0.0
        if d_active:
            gl, gm, gf = self._d1(recon_post, lens, True)
            gl2, gm2, gf2 = self._d2(recon_post, lens, True)
            l1 = O.lsgan_g_loss(gl, gm)
            l2 = torch.tensor(0.0)
            for i in range(len(gl2)):
                l2 = l2 + O.lsgan_g_loss(gl2[i], gm2[0])
            l2 = l2 / len(gl2)
            loss_gan = 0.5 * (l1 + l2)
            lam_g, lam_fm = t["loss_weights"]["Gloss_lambda"], t["loss_weights"]["fm_lambda"]
            if t["use_fm_loss"]:
                with torch.no_grad():
                    rf = self._d1(real, lens, True)[2]
                    rf2 = self._d2(real, lens, True)[2]
                f1 = torch.tensor(0.0)
                for (r, m), (f, _) in zip(rf, gf):
                    f1 = f1 + O.masked_mae(f, r, m)
                f1 = f1 / len(rf)
                f2 = torch.tensor(0.0)
                for i in range(len(gf2)):
                    for (r, m), (f, _) in zip(rf2[i], gf2[i]):
                        f2 = f2 + O.masked_mae(f, r, m)
                    f2 = f2 / len(rf2[i])            # QUIRK train.py:471-472: cumulative division inside the bin loop
                f2 = f2 / len(gf2)
                loss_fm = 0.5 * (f1 + f2)
        w = t["loss_weights"]
        total = (lp * w.get("recon_lambda_pre", 1.0) + lq * w.get("recon_lambda_post", 2.0)
                 + loss_gan * lam_g + loss_fm * lam_fm)
        total.backward()
        self.last_grads_g = self._snapshot(self.Pg)
        clip = t.get("clip_grad_norm", 1.0)
        if clip:
            self.last_norm_g = float(torch.nn.utils.clip_grad_norm_(self.g_params, clip))
        self.opt_g.step()
        self.sched_g.step()
        return {"loss_g_total": total.item(), "loss_recon_pre": lp.item(), "loss_recon_post": lq.item(),
                "loss_gan": float(loss_gan.detach()), "loss_fm": float(loss_fm.detach())}

    # -- train.py:516-537 ------------------------------------------------------
    def step(self, x, lens, d_active=True):
        self._gate_call = 0
        recon_pre, recon_post = O.preencoder_forward(self.Pg, x, lens, self.levels, self.gen_training, self.p_drop)
        loss_d = self.train_discriminator(x, recon_post, lens) if d_active else 0.0
        out = self.train_generator(x, recon_pre, recon_post, lens, d_active)
        out["loss_d"] = loss_d
        out["learning_rate"] = self.sched_g.get_last_lr()[0]
        out["recon_pre"], out["recon_post"] = recon_pre.detach(), recon_post.detach()
        return out


# ------------------------------------------------------------------------------- random init
def init_generator_params(cfg, seed=42) -> Dict[str, torch.Tensor]:
    """Random generator parameters with the reference's key names / shapes and
    PyTorch-default-like fan-in init (kaiming-uniform a=sqrt(5) == U(-1/sqrt(fan_in),
    1/sqrt(fan_in)) for weights and biases; weight-norm gain g = ||v||).  Used for
    the CPU baseline and for GPU-vs-oracle parity at sizes with no fixture."""
    g = torch.Generator().manual_seed(seed)
    m = cfg["model"]
    gc = m["generator"]
    mel, ch, ks = m["mel_channels"], gc["channels"], gc["kernel_sizes"]
    nq = len(gc["fsq_levels"])
    P: Dict[str, torch.Tensor] = {}

    def uni(shape, fan_in):
        b = 1.0 / (fan_in ** 0.5)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    def linear(name, cin, cout, bias=True):
        P[name + ".weight"] = uni((cout, cin), cin)
        if bias:
            P[name + ".bias"] = uni((cout,), cin)

    def wn_new(name, shape):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        v = uni(shape, fan_in)
        P[name + ".bias"] = uni((shape[0],), fan_in)
        P[name + ".parametrizations.weight.original0"] = v.flatten(1).norm(dim=1).view(-1, *([1] * (len(shape) - 1)))
        P[name + ".parametrizations.weight.original1"] = v

    def wn_old(name, shape):
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        v = uni(shape, fan_in)
        P[name + ".bias"] = uni((shape[0],), fan_in)
        P[name + ".weight_g"] = v.flatten(1).norm(dim=1).view(-1, *([1] * (len(shape) - 1)))
        P[name + ".weight_v"] = v

    def cb2d(name, c0):
        wn_new(name + ".dw", (1, 1, 5, 5))
        wn_new(name + ".pw", (c0, 1, 1, 1))
        P[name + ".conv_out.weight"] = uni((1, c0, 1, 1), c0)
        P[name + ".conv_out.bias"] = uni((1,), c0)

    linear("proj", mel, ch[0])
    cb2d("pre", ch[0])
    for i in range(len(ch) - 1):
        n = f"encoder_blocks.{i}"
        wn_new(n + ".conv1", (ch[i + 1], ch[i], ks[i]))
        wn_new(n + ".conv2", (ch[i + 1], ch[i + 1], ks[i]))
        c = ch[i + 1]
        linear(n + ".cbam.channel_attention.mlp.0", c, c // 8)
        linear(n + ".cbam.channel_attention.mlp.2", c // 8, c)
        P[n + ".cbam.spatial_attention.conv.weight"] = uni((1, 2, 7), 14)
        P[n + ".relu.beta"] = torch.tensor(1.0)
        P[n + ".relu.gamma"] = torch.tensor(0.5)
        if ch[i] != ch[i + 1]:
            P[n + ".residual.weight"] = uni((ch[i + 1], ch[i], 1), ch[i])
            P[n + ".residual.bias"] = uni((ch[i + 1],), ch[i])
    linear("q_in_proj", ch[-1], nq)
    linear("q_out_proj", nq, ch[-1])
    rch, rks = list(reversed(ch)), list(reversed(ks))
    for i in range(len(rch) - 1):
        n = f"decoder_blocks.{i}"
        wn_old(n + ".conv1", (rch[i + 1], rch[i], rks[i]))
        wn_old(n + ".conv2", (rch[i + 1], rch[i + 1], rks[i]))
        P[n + ".relu.beta"] = torch.tensor(1.0)
        P[n + ".relu.gamma"] = torch.tensor(0.5)
        if rch[i] != rch[i + 1]:
            P[n + ".residual.weight"] = uni((rch[i + 1], rch[i], 1), rch[i])
            P[n + ".residual.bias"] = uni((rch[i + 1],), rch[i])
    cb2d("post", ch[0])
    linear("out_proj", ch[0], mel)
    hid = mel // gc.get("refiner_hidden_proj_divisor", 8)
    linear("hidden_proj", ch[0], hid)
    base, depth = gc.get("refiner_base_channels", 128), gc.get("refiner_depth", 3)
    chs = [base * 2 ** i for i in range(depth + 1)]

    def cblock(name, cin, cout):
        wn_new(name + ".conv1", (cout, cin, 3, 3))
        wn_new(name + ".conv2", (cout, cout, 3, 3))

    cblock("refiner.pre", 1, chs[0])
    for i in range(depth):
        cblock(f"refiner.downs.{i}.conv", chs[i], chs[i + 1])
    cblock("refiner.mid", chs[-1], chs[-1])
    for i in range(depth):
        cblock(f"refiner.ups.{i}.conv", chs[depth - i] + chs[depth - i - 1], chs[depth - i - 1])
    wn_new("refiner.post", (1, chs[0], 3, 3))
    linear("refiner.reproj", mel + hid, mel, bias=False)
    return P


def init_patch_params(mel, hidden, kernels, seed=1, prefix="") -> Dict[str, torch.Tensor]:
    """discriminators.py:99-181: N(0, 0.02) conv weights, zero bias, u/v ~ normalised N(0,1)."""
    g = torch.Generator().manual_seed(seed)
    P: Dict[str, torch.Tensor] = {}
    cin = 1
    chans = list(hidden) + [1]
    for i, (cout, k) in enumerate(zip(chans, kernels)):
        kh, kw = (k, k) if isinstance(k, int) else k
        w = torch.randn((cout, cin, kh, kw), generator=g) * 0.02
        P[f"{prefix}convs.{i}.bias"] = torch.zeros(cout)
        P[f"{prefix}convs.{i}.weight_orig"] = w
        P[f"{prefix}convs.{i}.weight_u"] = torch.nn.functional.normalize(torch.randn(cout, generator=g), dim=0, eps=1e-12)
        P[f"{prefix}convs.{i}.weight_v"] = torch.nn.functional.normalize(torch.randn(cin * kh * kw, generator=g), dim=0,
                                                                         eps=1e-12)
        cin = cout
    c = hidden[-1]
    r = max(1, c // 8)
    b = 1.0 / c ** 0.5
    P[prefix + "se_block.fc1.weight"] = (torch.rand((r, c), generator=g) * 2 - 1) * b
    P[prefix + "se_block.fc1.bias"] = (torch.rand((r,), generator=g) * 2 - 1) * b
    b = 1.0 / r ** 0.5
    P[prefix + "se_block.fc2.weight"] = (torch.rand((c, r), generator=g) * 2 - 1) * b
    P[prefix + "se_block.fc2.bias"] = (torch.rand((c,), generator=g) * 2 - 1) * b
    return P


def init_multibin_params(mel, n_bins, hidden, kernels, seed=2) -> Dict[str, torch.Tensor]:
    P: Dict[str, torch.Tensor] = {}
    for j in range(n_bins):
        P.update(init_patch_params(mel // n_bins, hidden, [(3, k) for k in kernels], seed=seed * 100 + j,
                                   prefix=f"discriminators.{j}."))
    return P


def make_leaves(P: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in P.items():
        v = v.clone()
        if v.is_floating_point() and not O.is_sn_buffer(k, P):
            v.requires_grad_(True)
        out[k] = v
    return out
