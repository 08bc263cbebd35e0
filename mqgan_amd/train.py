"""Trainer for the PreEncoder VQ-GAN with the reference's configuration schema, step semantics,
logged scalar names and checkpoint layout (reference train.py:87-197 data, :202-652 Trainer),
running on the HIP kernels, one process per GPU.

What differs from the reference, on purpose:
  * no host synchronisation inside the step: losses stay on the device and are only read when a log
    line is due (the reference calls .item() >= 6 times per step, train.py:410, 496-500);
  * parameters, gradients and Adam moments live in flat arenas: clip_grad_norm_ + Adam are two kernel
    launches per model, and the data-parallel all-reduce works on a few large buckets (ddp.py);
  * in the generator step the discriminators run with detached weights: the reference computes
    discriminator weight-gradients there that it never uses (they are zeroed at train.py:381 before
    the next discriminator step), so results are unaffected;
  * optional extra checkpoint keys (discriminators, spectral-norm u/v, LeCam EMA, scheduler step) that
    the reference loader ignores.
"""
import glob
import json
import os
import random
import time
from functools import partial

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .comm import Comm, default_comm
from .ddp import FlatArena, GradReducer, broadcast_arena
from .discriminators import MelSpectrogramPatchDiscriminator2D, MultiBinDiscriminator, run_discriminators
from .losses import LSGANLoss, masked_mae_deferred, recon_loss_deferred
from .nn_base import share_effective_weights
from .preencoder import PreEncoder


# ----------------------------------------------------------------------------- data (train.py:87-197)
class RealMelSpectrogramDataset(torch.utils.data.Dataset):
    def __init__(self, data_dir: str, crop_len=None):
        self.real_dir, self.crop_len = data_dir, crop_len
        if not os.path.isdir(data_dir):
            raise FileNotFoundError(f"Directory not found: {data_dir}")
        self.filenames = [os.path.join(root, fn) for root, _, files in os.walk(data_dir) for fn in files if fn.endswith(".npy")]

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, idx):
        path = self.filenames[idx]
        try:
            mel = np.load(path)
        except Exception as e:          # bad file -> None -> dropped by the collate (train.py:110-118)
            print(f"[Load error] {path}: {e}")
            return None
        if mel.ndim != 2:
            print(f"[Shape error] {path}: shape={mel.shape}")
            return None
        full_len, target = mel.shape[0], self.crop_len
        if target is not None:
            if full_len > target:
                start = np.random.randint(0, full_len - target + 1)
                mel = mel[start:start + target]
            elif full_len < target:
                mel = np.concatenate([mel, np.zeros((target - full_len, mel.shape[1]), dtype=mel.dtype)], axis=0)
            mel = mel[:target]
        mel_len = min(full_len, target) if target is not None else full_len
        return torch.as_tensor(mel.astype(np.float32)), int(mel_len), os.path.basename(path)


def pad_collate_fn(batch, crop_lens=None):
    """train.py:140-197: drop failed loads; one crop length per batch (int, or random pick from a list);
    random crop longer items, zero right-pad shorter ones; lengths as int32."""
    batch = [item for item in batch if item is not None]
    if not batch:
        return None, None, None
    specs, lens, names = zip(*batch)
    tgt = None
    if crop_lens is not None:
        tgt = int(random.choice(crop_lens)) if isinstance(crop_lens, (list, tuple)) else int(crop_lens)
    if tgt is None and len({m.shape[0] for m in specs}) == 1:
        return torch.stack([torch.as_tensor(m, dtype=torch.float32) for m in specs]), torch.tensor(lens, dtype=torch.int32), names
    if tgt is None:
        tgt = max(lens)
    out, new_lens = [], []
    for mel, full in zip(specs, lens):
        mel = torch.as_tensor(mel, dtype=torch.float32)
        if full > tgt:
            start = random.randint(0, full - tgt)
            mel = mel[start:start + tgt]
        elif full < tgt:
            mel = torch.nn.functional.pad(mel, (0, 0, 0, tgt - full))
        out.append(mel)
        new_lens.append(min(full, tgt))
    return torch.stack(out), torch.tensor(new_lens, dtype=torch.int32), names


def synthetic_mel(T: int, mel: int, index: int) -> np.ndarray:
    """Synthetic log-mel of the survey's measurement plan."""
    rng = np.random.default_rng(1234 + index)
    return np.clip(rng.standard_normal((T, mel)) * 2.0 - 4.0, -11.5129, 5.0).astype(np.float32)


# ----------------------------------------------------------------------------- optimiser on a flat arena
class FlatAdam:
    """torch.optim.Adam semantics (no weight decay / amsgrad) over a FlatArena, with clip_grad_norm_
    fused in.  state_dict()/load_state_dict() speak torch.optim.Adam's format, so the reference's
    checkpoints (optimizer_g_state_dict / optimizer_d_state_dict) round-trip."""

    def __init__(self, arena: FlatArena, lr, betas=(0.9, 0.999), eps=1e-8):
        self.arena, self.base_lr, self.lr, self.betas, self.eps = arena, lr, lr, tuple(betas), eps
        self.exp_avg = torch.zeros_like(arena.flat_param)
        self.exp_avg_sq = torch.zeros_like(arena.flat_param)
        self.sumsq = torch.zeros(1, device=arena.flat_param.device)
        self.step_count = 0

    def zero_grad(self):
        self.arena.zero_grad()

    def step(self, max_norm=0.0, grad_scale=1.0):
        self.step_count += 1
        self.arena.gather()                    # gather-mode arenas: whatever the reducer has not moved yet
        if max_norm:
            ops.sumsq(self.arena.flat_grad, self.sumsq)
        ops.adam_step(self.arena.flat_param, self.arena.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.step_count, self.sumsq if max_norm else None, max_norm or 0.0, grad_scale)

    def grad_norm(self, grad_scale=1.0):
        """Pre-clip total norm of the last step (device tensor; reading it synchronises)."""
        return self.sumsq.sqrt() * grad_scale

    def state_dict(self):
        state = {}
        for i, (p, o) in enumerate(zip(self.arena.params, self.arena.offsets)):
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[o:o + p.numel()].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + p.numel()].view_as(p).clone()}
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "initial_lr": self.base_lr,
                 "params": list(range(len(self.arena.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        for i, st in sd.get("state", {}).items():
            i = int(i)
            p, o = self.arena.params[i], self.arena.offsets[i]
            self.exp_avg[o:o + p.numel()].view_as(p).copy_(st["exp_avg"])
            self.exp_avg_sq[o:o + p.numel()].view_as(p).copy_(st["exp_avg_sq"])
            self.step_count = max(self.step_count, int(float(st["step"])))
        if sd.get("param_groups"):
            self.lr = sd["param_groups"][0].get("lr", self.lr)


class WarmupLR:
    """LambdaLR(lr_lambda = min((step + 1) / warmup, 1)) on the generator optimiser (train.py:326-329)."""

    def __init__(self, opt: FlatAdam, warmup_steps: int):
        self.opt, self.warmup, self.last_epoch = opt, warmup_steps, 0
        opt.lr = opt.base_lr * min(1.0 / warmup_steps, 1.0)

    def step(self):
        self.last_epoch += 1
        self.opt.lr = self.opt.base_lr * min((self.last_epoch + 1) / self.warmup, 1.0)

    def get_last_lr(self):
        return [self.opt.lr]


# ----------------------------------------------------------------------------- trainer
class Trainer:
    def __init__(self, config, device=None, build_data=True, log_fn=None, comm: Comm = None, force_collectives: bool = False):
        self.config = config
        # data-parallel communicator: torch.distributed when a process group is up, else the single-process identity.
        # force_collectives (needs an initialised process group): every collective of the data-parallel step is issued
        # even with one rank - the gradient hooks, bucket all-reduces on the group's side stream and the loss-sum
        # all-reduces then run exactly as with eight ranks (tests/test_gpu_rccl.py, bench.py --force-collectives)
        if force_collectives and comm is None:
            from .comm import TorchComm
            comm = TorchComm(force=True)
        self.force_collectives = bool(force_collectives)
        self.comm = comm if comm is not None else default_comm()
        self.rank, self.world = self.comm.rank, self.comm.world
        self.device = torch.device(device) if device is not None else self._get_device()
        if self.device.type == "cuda":
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            torch.cuda.set_device(self.device)           # kernels launch on the current device's stream (_lib.stream)
        self.start_epoch = 1
        self.epoch = 1
        self._setup_seeds()
        self.train_loader = self.eval_dataset = None
        if build_data:
            self.train_loader, self.eval_dataset = self._get_dataloaders()
        self.generator, self.patch_discriminator, self.multibin_discriminator = self._init_models()
        # gradients stay where the backward kernels wrote them; one launch per step (per all-reduce bucket) moves them
        # into the arenas instead of ~350 small accumulation kernels
        self.arena_g = FlatArena([self.generator], gather=True)
        self.arena_d = FlatArena([self.patch_discriminator, self.multibin_discriminator], gather=True)
        broadcast_arena(self.arena_g)
        broadcast_arena(self.arena_d)
        self.optimizer_g, self.optimizer_d = self._init_optimizers()
        self.scheduler_g = WarmupLR(self.optimizer_g, config["training"]["warmup_steps"])
        self.reducer_g = GradReducer(self.arena_g, comm=self.comm, force=self.force_collectives)
        self.reducer_d = GradReducer(self.arena_d, comm=self.comm, force=self.force_collectives)
        self.gan_loss = LSGANLoss(comm=self.comm).to(self.device)   # ONE instance for all nine discriminators (train.py:216)
        # The reference picks bf16 autocast whenever the GPU supports it (train.py:220, 523); here the engine is an
        # explicit knob: training.precision = "fp32" (default; the 1e-4 parity path, BASELINE config 2), "fp32x3" or "bf16"
        # (bf16 MFMA operands in every conv / linear, f32 accumulation, f32 tensors; BASELINE config 3).
        prec = str(config["training"].get("precision", "fp32")).lower()
        if prec not in ("fp32", "f32", "bf16", "f32x3", "fp32x3"):
            raise ValueError(f"training.precision must be fp32, fp32x3 or bf16, got {prec!r}")
        self.use_bf16 = prec == "bf16"
        # bf16 mode stores the refiner's and the discriminators' activations as bf16 tensors, as the reference's autocast
        # region does (train.py:220, 523); training.bf16_storage: false keeps f32 tensors with in-kernel rounding
        self.bf16_storage = bool(config["training"].get("bf16_storage", True))
        # "fp32x3": f32 results from the bf16 matrix pipe (exact three-term split of every operand, 6 cross products);
        # opt-in - it has the error of an f32 FMA chain but is not bit-identical to the default engine
        self.conv_precision = "bf16" if self.use_bf16 else ("f32x3" if prec.endswith("x3") else "f32")
        self.log_fn = log_fn
        self.global_step = 0
        self._init_checkpoint_handling()

    # -- setup --------------------------------------------------------------------------------
    def _get_device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("mqgan_amd.Trainer needs a HIP device (no CPU path); use oracle/ for CPU runs")
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))

    def _setup_seeds(self):
        seed = self.config["training"]["seed"]
        random.seed(seed + self.rank)
        np.random.seed(seed + self.rank)
        torch.manual_seed(seed)                          # identical initial weights on every rank
        ops.manual_seed(seed, self.rank)                 # per-rank dropout streams

    def _get_dataloaders(self):
        cfg = self.config["data"]
        full = RealMelSpectrogramDataset(cfg["data_dir"], None)
        if len(full) == 0:
            raise ValueError("Dataset is empty after initialization.")
        eval_size = int(cfg["validation_split"] * len(full))
        train_size = len(full) - eval_size
        gen = torch.Generator().manual_seed(self.config["training"]["seed"])
        train_ds, eval_ds = torch.utils.data.random_split(full, [train_size, eval_size], generator=gen)
        sampler = None
        if self.world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(train_ds, self.world, self.rank, shuffle=True,
                                                                      seed=self.config["training"]["seed"])
        # data parallel: every rank must take the same number of equally sized steps (the collectives are matched one
        # to one), so the ragged last batch is dropped there; a single process keeps it like the reference does
        loader = torch.utils.data.DataLoader(train_ds, batch_size=cfg["batch_size"], shuffle=sampler is None, sampler=sampler,
                                             num_workers=cfg.get("num_workers", 0), pin_memory=True, drop_last=self.world > 1,
                                             collate_fn=partial(pad_collate_fn, crop_lens=cfg["crop_len"]))
        return loader, eval_ds

    def _init_models(self):
        m = self.config["model"]
        g = m["generator"]
        gen = PreEncoder(mel_channels=m["mel_channels"], channels=g["channels"], kernel_sizes=g["kernel_sizes"],
                         dropout=g["dropout"], fsq_levels=g["fsq_levels"],
                         refiner_base_channels=g.get("refiner_base_channels", 128), refiner_depth=g.get("refiner_depth", 3),
                         refiner_hidden_proj_divisor=g.get("refiner_hidden_proj_divisor", 8)).to(self.device)
        p = m["discriminator_patch"]
        d1 = MelSpectrogramPatchDiscriminator2D(m["mel_channels"], hidden_channels=p["hidden_channels"],
                                                kernel_sizes=p["kernel_sizes"], stride=p["strides"]).to(self.device)
        b = m["discriminator_multibin"]
        d2 = MultiBinDiscriminator(m["mel_channels"], hidden_channels=b["hidden_channels"], kernel_sizes=b["kernel_sizes"],
                                   n_bins=b["n_bins"], n_no_strides=b["n_no_strides"]).to(self.device)
        return gen, d1, d2

    def _init_optimizers(self):
        t = self.config["training"]
        og = FlatAdam(self.arena_g, t["lr"], (t["beta1"], t["beta2"]))
        od = FlatAdam(self.arena_d, t["lr"] * t["lr_d_factor"], (t["d_beta1"], t["d_beta2"]))
        return og, od

    # -- checkpoints (train.py:339-378, 625-636) -------------------------------------------------
    def _init_checkpoint_handling(self):
        out = self.config.get("data", {}).get("output_dir")
        latest = None
        if out:
            latest = max(glob.glob(os.path.join(out, "checkpoint_epoch_*.pth")), key=os.path.getctime, default=None)
        if latest:
            self._load_full_checkpoint(latest)
        else:
            self._load_pretrained_generator()

    def _load_full_checkpoint(self, path):
        ck = torch.load(path, map_location=self.device, weights_only=False)
        self.generator.load_state_dict(ck["model_state_dict"])
        self.optimizer_g.load_state_dict(ck["optimizer_g_state_dict"])
        self.optimizer_d.load_state_dict(ck["optimizer_d_state_dict"])
        extra = ck.get("mqgan_amd_extra")                # optional keys the reference ignores
        if extra:
            self.patch_discriminator.load_state_dict(extra["patch_discriminator"])
            self.multibin_discriminator.load_state_dict(extra["multibin_discriminator"])
            self.gan_loss._state.copy_(extra["gan_loss_state"])
            self.scheduler_g.last_epoch = extra["scheduler_last_epoch"] - 1
            self.scheduler_g.step()
        self.arena_g.rebind()
        self.arena_d.rebind()
        self.start_epoch = ck["epoch"] + 1

    def _load_pretrained_generator(self):
        path = self.config["training"].get("pretrained")
        if path and os.path.isfile(path):
            ck = torch.load(path, map_location=self.device, weights_only=False)
            sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
            self.generator.load_state_dict({k.replace("module.", ""): v for k, v in sd.items()}, strict=False)
            self.arena_g.rebind()

    def _save_checkpoint(self, epoch):
        if self.rank != 0:
            return None
        out = self.config["data"]["output_dir"]
        os.makedirs(out, exist_ok=True)
        path = os.path.join(out, f"checkpoint_epoch_{epoch:03d}.pth")
        torch.save({
            "epoch": epoch,
            "model_state_dict": self.generator.state_dict(),
            "optimizer_g_state_dict": self.optimizer_g.state_dict(),
            "optimizer_d_state_dict": self.optimizer_d.state_dict(),
            "scaler_g_state_dict": {},                    # GradScaler is disabled under bf16/fp32 (train.py:223-224)
            "scaler_d_state_dict": {},
            "config": self.config,
            "mqgan_amd_extra": {"patch_discriminator": self.patch_discriminator.state_dict(),
                                "multibin_discriminator": self.multibin_discriminator.state_dict(),
                                "gan_loss_state": self.gan_loss._state.clone(),
                                "scheduler_last_epoch": self.scheduler_g.last_epoch},
        }, path)
        return path

    # -- discriminator passes -------------------------------------------------------------------
    def _d1(self, x, lens, feats=False, detach_params=False):
        return run_discriminators([self.patch_discriminator], x, lens, feats, detach_params=detach_params, pregate=True)

    def _d2(self, x, lens, feats=False, detach_params=False):
        return run_discriminators(list(self.multibin_discriminator.discriminators), x, lens, feats, detach_params=detach_params,
                                  pregate=True)

    def _bands(self, t):
        G = self.multibin_discriminator.n_bins
        B = t.shape[0] // G
        return [t[j * B:(j + 1) * B] for j in range(G)]

    # -- train.py:380-410 -----------------------------------------------------------------------
    def _train_discriminator(self, real, recon, lens):
        self.optimizer_d.zero_grad()
        self.reducer_d.start()
        fake = recon.detach()
        B = real.shape[0]
        G = self.multibin_discriminator.n_bins
        if not (self.patch_discriminator.training or self.multibin_discriminator.training):
            # Eval-mode discriminators (their state from the first generator step of an epoch on, train.py:415-416) are
            # per-sample functions with fixed weights: real and fake go through as ONE batch of 2B - half the launches,
            # fuller tiles, and every weight gradient is formed by one GEMM instead of two plus an add.
            both, lens2 = torch.cat([real, fake], 0), torch.cat([lens, lens], 0)
            al, amask, _ = self._d1(both, lens2)
            rmask = amask[:B]
            first = (al[:B], al[B:], rmask, rmask)
            al2, amask2, _ = self._d2(both, lens2)       # band j owns rows [2jB, 2(j+1)B): real first, then fake
            m0 = amask2[:B].contiguous()                 # the reference uses the mask of bin 0 for every bin (:392-393)
            pairs = [(al2[2 * j * B:(2 * j + 1) * B], al2[(2 * j + 1) * B:(2 * j + 2) * B]) for j in range(G)]
        else:
            # train mode: every forward runs one power iteration (u, v change between passes), so the passes stay separate
            rl, rmask, _ = self._d1(real, lens)
            fl, _, _ = self._d1(fake, lens)
            first = (rl, fl, rmask, rmask)
            rl2, rmask2, _ = self._d2(real, lens)
            fl2, _, _ = self._d2(fake, lens)
            m0 = rmask2[:B].contiguous()
            pairs = list(zip(self._bands(rl2), self._bands(fl2)))
        # nine LSGANLoss.discriminator_loss calls in the reference's order (patch D, then bins 0..7: one EMA chain,
        # train.py:385-396).  Their numerators and counts are summed over the data-parallel ranks before the division
        # and before the EMA update: global normalisers, replica-consistent LeCam state, exact for ragged shards.
        losses = self.gan_loss.discriminator_losses([first] + [(r, f, m0, m0) for r, f in pairs])
        loss = losses[0]
        mbd = None
        for l in losses[1:]:
            mbd = l if mbd is None else ops.add(mbd, l)
        loss_d = ops.add(loss, mbd, 1.0, 1.0 / G)
        loss_d.backward()
        self.reducer_d.finish()
        clip = self.config["training"].get("clip_grad_norm", 1.0)
        # every loss is normalised by GLOBAL counts, so the SUM of the ranks' gradients is the global-batch gradient
        self.optimizer_d.step(max_norm=clip or 0.0, grad_scale=1.0)
        return loss_d.detach()

    # -- train.py:412-501 -----------------------------------------------------------------------
    def _train_generator(self, real, recon_pre, recon_post, lens):
        t = self.config["training"]
        self.optimizer_g.zero_grad()
        self.reducer_g.start()
        self.patch_discriminator.eval()                  # never undone until the next epoch (:415-416 vs :505-506)
        self.multibin_discriminator.eval()
        # Every loss of this step is a ratio of sums over the batch: all their numerators / counts go into ONE buffer and cross
        # the data-parallel ranks in ONE all-reduce (ops.SumsBatch) - the *_deferred calls launch the sums kernels now and
        # return the function that forms the loss once the buffer has been reduced.
        sums = ops.SumsBatch(real, self.comm)
        loss_pre_f = recon_loss_deferred(recon_pre, real, lens, sums)     # all + 0.25 * group16 (:419-425)
        loss_post_f = recon_loss_deferred(recon_post, real, lens, sums)
        d_active = self.epoch >= t["discriminator_train_start_epoch"]
        zero = torch.zeros((), device=self.device)
        loss_gan, loss_fm = zero, zero
        lam_g = lam_fm = 0.0
        use_fm = d_active and bool(t["use_fm_loss"])
        if d_active:
            with share_effective_weights():              # the FM passes over `real` reuse the weight preparations
                gl, gmask, gf = self._d1(recon_post, lens, feats=use_fm, detach_params=True)
                gl2, gmask2, gf2 = self._d2(recon_post, lens, feats=use_fm, detach_params=True)
                if use_fm:
                    with torch.no_grad():
                        _, _, rf = self._d1(real, lens, feats=True, detach_params=True)
                        _, _, rf2 = self._d2(real, lens, feats=True, detach_params=True)
            B = real.shape[0]
            G = self.multibin_discriminator.n_bins
            m0 = gmask2[:B].contiguous()
            l1_f = self.gan_loss.generator_loss_deferred(gl, gmask, sums)
            l2_f = [self.gan_loss.generator_loss_deferred(g_, m0, sums) for g_ in self._bands(gl2)]
            lam_g, lam_fm = t["loss_weights"]["Gloss_lambda"], t["loss_weights"]["fm_lambda"]
            if use_fm:
                # gs: the feature's layer takes a pre-activation gradient
                f1_f = [masked_mae_deferred(f, r, m, sums, leaky_slope=gs) for (r, m, _), (f, _, gs) in zip(rf, gf)]
                # multi-bin features: one loss node per layer covers all bands (one gradient tensor per feature instead of a
                # zero-filled slice gradient per band)
                f2_f = [masked_mae_deferred(f, r, m[:r.shape[0] // G].contiguous(), sums, leaky_slope=gs, n_bands=G)
                        for (r, m, _), (f, _, gs) in zip(rf2, gf2)]
        sums.reduce()
        loss_pre, loss_post = loss_pre_f(), loss_post_f()
        if d_active:
            l1 = l1_f()
            l2 = None
            for lf in l2_f:
                l = lf()
                l2 = l if l2 is None else ops.add(l2, l)
            loss_gan = ops.add(l1, l2, 0.5, 0.5 / G)
            if use_fm:
                f1 = None
                for lf in f1_f:
                    l = lf()
                    f1 = l if f1 is None else ops.add(f1, l)
                f1 = ops.scale(f1, 1.0 / len(rf))
                # per band j the sum over the feature layers, then the reference's cumulative division inside the bin loop
                # (train.py:466-472): f2 = sum_j S_j / nf^(G-j) / G
                nf = len(rf2)
                S = None
                for lf in f2_f:
                    lb = lf()
                    S = lb if S is None else S + lb
                if getattr(self, "_fm_band_w", None) is None or self._fm_band_w[0] != (nf, G):
                    self._fm_band_w = ((nf, G), torch.tensor([1.0 / (float(nf) ** (G - j)) / G for j in range(G)],
                                                             dtype=torch.float32, device=self.device))
                f2 = (S * self._fm_band_w[1]).sum()
                loss_fm = ops.add(f1, f2, 0.5, 0.5)
        w = t["loss_weights"]
        total = ops.add(loss_pre, loss_post, w.get("recon_lambda_pre", 1.0), w.get("recon_lambda_post", 2.0))
        if d_active:
            total = ops.add(total, loss_gan, 1.0, lam_g)
            if t["use_fm_loss"]:
                total = ops.add(total, loss_fm, 1.0, lam_fm)
        total.backward()
        self.reducer_g.finish()
        clip = t.get("clip_grad_norm", 1.0)
        self.optimizer_g.step(max_norm=clip or 0.0, grad_scale=1.0)       # sum of the ranks' gradients (global normalisers)
        self.scheduler_g.step()
        return {"loss_g_total": total.detach(), "loss_recon_pre": loss_pre.detach(), "loss_recon_post": loss_post.detach(),
                "loss_gan": loss_gan.detach(), "loss_fm": loss_fm.detach()}

    # -- one optimisation step on a device batch (train.py:516-537) --------------------------------
    def train_step(self, real, lens):
        """real (B, T, mel) fp32 and lens (B,) int32, both on the device -> dict of device scalars."""
        with ops.conv_precision(self.conv_precision), ops.request_act_storage(self.use_bf16 and self.bf16_storage):
            recon_pre, recon_post = self.generator(real, lens)
            out = {"loss_d": torch.zeros((), device=self.device)}
            if self.epoch >= self.config["training"]["discriminator_train_start_epoch"]:
                out["loss_d"] = self._train_discriminator(real, recon_post, lens)
            out.update(self._train_generator(real, recon_pre, recon_post, lens))
        out["learning_rate"] = self.scheduler_g.get_last_lr()[0]
        self.global_step += 1
        return out

    def _train_epoch(self, epoch, next_epoch=...):
        self.generator.train()
        self.patch_discriminator.train()
        self.multibin_discriminator.train()
        self.epoch = epoch
        log_every = self.config.get("logging", {}).get("log_interval", 50)
        frames, t0 = torch.zeros((), dtype=torch.int64, device=self.device), time.time()
        from .data import PrefetchLoader
        if getattr(self, "_prefetch", None) is None or self._prefetch.loader is not self.train_loader:
            self._prefetch = PrefetchLoader(self.train_loader, self.device, depth=2)      # lives across epochs (mqgan_amd/data.py)
        # the sampler's shuffle (DistributedSampler.set_epoch) is set by the producer thread of each epoch; the next epoch's first
        # batches are staged while this epoch's last steps run
        last = self.config["training"].get("num_epochs", epoch)
        nxt = (epoch + 1 if epoch < last else None) if next_epoch is Ellipsis else next_epoch
        for i, (real, lens, _) in enumerate(self._prefetch.epoch(epoch, next_epoch=nxt)):
            empty = real is None or real.size(0) == 0   # every file of the batch failed to load (train.py:512-514)
            if self.comm.active:
                # the step is full of collectives: either every rank takes it or none does.  Agreed on the HOST (gloo twin
                # of the RCCL group): no device synchronisation, the host keeps running ahead of the stream
                empty = self.comm.any_rank(empty)
            if empty:
                continue
            out = self.train_step(real, lens)
            frames += lens.sum()                        # the metric counts VALID frames (sum of lengths); no host sync
            if self.log_fn is not None and (i + 1) % log_every == 0:
                rec = {k: (float(v) if torch.is_tensor(v) else v) for k, v in out.items()}
                rec.update(epoch=epoch, step=self.global_step, frames_per_s=int(frames.item()) / (time.time() - t0))
                self.log_fn(rec)

    def _evaluate(self, epoch):
        """train.py:575-623 in numbers: the reference reconstructs ``num_plot_examples`` random utterances of the validation split
        (whole files, generator in eval mode, no dropout) and logs pictures of them to wandb; the pictures are out of scope
        (SURVEY 2), what they are looked at for is not - so the same samples are reconstructed the same way and the masked
        reconstruction losses of both outputs (the training loss's terms, train.py:419-425) go to the log:
        ``eval_loss_recon_pre / eval_loss_recon_post`` (mean over the samples), ``eval_frames``, ``eval_samples``.  Rank 0 only
        (every rank holds the same weights); nothing is logged when the validation split is empty."""
        from .losses import recon_loss
        ds = self.eval_dataset
        if ds is None or len(ds) == 0 or self.rank != 0:
            return None
        n = min(self.config.get("logging", {}).get("num_plot_examples", 3), len(ds))
        was_training = self.generator.training
        self.generator.eval()
        pre, post, frames, used = 0.0, 0.0, 0, 0
        try:
            with torch.no_grad(), ops.conv_precision(self.conv_precision), ops.request_act_storage(self.use_bf16 and self.bf16_storage):
                for idx in random.sample(range(len(ds)), n):
                    item = ds.dataset[ds.indices[idx]]
                    if item is None:                       # unreadable file: skipped, as in training
                        continue
                    mel, mel_len, _ = item
                    x = mel.to(self.device).unsqueeze(0)
                    lens = torch.tensor([mel_len], dtype=torch.int32, device=self.device)
                    recon_pre, recon_post = self.generator(x, lens)
                    pre += float(recon_loss(recon_pre, x, lens))
                    post += float(recon_loss(recon_post, x, lens))
                    frames += int(mel_len)
                    used += 1
        finally:
            self.generator.train(was_training)
        if used == 0:
            return None
        rec = {"epoch": epoch, "step": self.global_step, "eval_loss_recon_pre": pre / used, "eval_loss_recon_post": post / used,
               "eval_samples": used, "eval_frames": frames}
        if self.log_fn is not None:
            self.log_fn(rec)
        return rec

    def train(self):
        last = self.config["training"]["num_epochs"]
        eval_every = self.config.get("logging", {}).get("eval_interval", 0)
        for epoch in range(self.start_epoch, last + 1):
            self._train_epoch(epoch, next_epoch=epoch + 1 if epoch < last else None)
            if eval_every and epoch % eval_every == 0:     # train.py:650-651
                self._evaluate(epoch)
            if epoch % self.config["logging"]["save_interval"] == 0:
                self._save_checkpoint(epoch)
        if getattr(self, "_prefetch", None) is not None:
            self._prefetch.close()


def jsonl_logger(path):
    def log(rec):
        with open(path, "a") as f:
            f.write(json.dumps(rec) + "\n")
    return log


def main(argv=None):
    import argparse
    import yaml
    ap = argparse.ArgumentParser(description="Train an MQGAN PreEncoder on MI355X (drop-in for the reference's train.py).")
    ap.add_argument("--config", type=str, default="config.yaml")
    ap.add_argument("--pretrained", type=str, default=None)
    ap.add_argument("--output_dir", type=str, default=None)
    args = ap.parse_args(argv)
    with open(args.config) as f:
        config = yaml.safe_load(f)
    if args.pretrained:
        config["training"]["pretrained"] = args.pretrained
    if args.output_dir:
        config["data"]["output_dir"] = args.output_dir
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    os.makedirs(config["data"]["output_dir"], exist_ok=True)
    tr = Trainer(config, log_fn=jsonl_logger(os.path.join(config["data"]["output_dir"], "train_log.jsonl")))
    tr.train()


if __name__ == "__main__":
    main()
