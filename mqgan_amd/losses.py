"""LSGANLoss (+ LeCam) and MaskedMelLoss with the reference's API (reference losses.py:5-116,
121-182) and masked_mae (train.py:38-45), on the HIP kernels.  No host synchronisation: the
reference's ``valid.item() > 0`` branch (losses.py:29) is taken on the device.

Every loss is a ratio of sums over the batch.  The modules / functions take an optional ``comm``
(mqgan_amd.comm): with more than one data-parallel rank the numerators and counts are all-reduced before
the division, so the normalisers are those of the GLOBAL batch - every rank then holds the
global loss value and the SUM of the ranks' gradients is the single-device gradient, ragged shards included.
"""
import torch
import torch.nn as nn

from . import ops
from .comm import Comm


def _padded_map(mask, logits):
    """Loss mask of the reference (True / non-zero = VALID, broadcastable to the logits (B, 1, H, W)) -> (uint8
    PADDED map, mask_mode) for the kernels: mode 0 = frame mask (B, W) - the uint8 masks the discriminators
    hand out, already 1 = padded -; mode 1 = one byte per element (B, H, W); (None, 0) = everything valid."""
    if mask is None:
        return None, 0
    B, H, W = logits.shape
    if mask.dtype == torch.uint8 and mask.dim() == 2:
        if tuple(mask.shape) != (B, W):
            raise ValueError(f"frame mask {tuple(mask.shape)} does not match logits (B, W) = {(B, W)}")
        return mask.contiguous(), 0
    m = mask
    if m.dim() == 4:
        m = m.reshape(m.shape[0], m.shape[2], m.shape[3]) if m.shape[1] == 1 else m.reshape(m.shape[0], m.shape[1], m.shape[2])
    if m.dim() != 3:
        raise ValueError(f"mask of shape {tuple(mask.shape)} cannot be broadcast to logits {(B, 1, H, W)}")
    valid = m != 0
    return (~valid).expand(B, H, W).to(torch.uint8).contiguous(), 1


def _bhw(logits):
    """(B, 1, H, W) or channels-last (B, H, W, 1) single-channel logits -> contiguous (B, H, W)."""
    if logits.dim() != 4:
        raise ValueError(f"expected 4-D single-channel logits, got {tuple(logits.shape)}")
    if logits.shape[1] == 1 and logits.shape[3] != 1:
        return logits.reshape(logits.shape[0], logits.shape[2], logits.shape[3]).contiguous()
    return logits.reshape(logits.shape[0], logits.shape[1], logits.shape[2]).contiguous()


class LSGANLoss(nn.Module):
    def __init__(self, real_label=1.0, fake_label=0.0, decay=0.99, use_lecam=True, comm: Comm = None):
        super().__init__()
        if real_label != 1.0 or fake_label != 0.0 or not use_lecam:
            raise NotImplementedError("the HIP LSGAN kernel implements labels (1, 0) with LeCam, as train.py uses it")
        self.real_label, self.fake_label, self.decay, self.use_lecam = real_label, fake_label, decay, use_lecam
        self.comm = comm
        # [ema_real, ema_fake, initialised, unused]; ONE state shared by every discriminator (train.py:216)
        self.register_buffer("_state", torch.zeros(4), persistent=False)

    @property
    def ema_real(self):
        return self._state[0]

    @property
    def ema_fake(self):
        return self._state[1]

    @property
    def ema_initialized(self):
        return bool(self._state[2].item() != 0)

    def discriminator_losses(self, calls):
        """n consecutive ``discriminator_loss`` calls as one unit: ``calls`` = [(real_output, fake_output, real_mask,
        fake_mask)].  Same results as calling discriminator_loss n times in this order (the EMA chain is followed call
        by call), but the sums of all calls travel in two small all-reduces instead of 2n."""
        pairs = []
        for real_output, fake_output, real_mask, fake_mask in calls:
            r, f = _bhw(real_output), _bhw(fake_output)
            rp, mr = _padded_map(real_mask, r)
            fp, mf = _padded_map(fake_mask, f)
            if mr != mf:                                # one mode per launch: lift the frame mask to a per-element map
                if mr == 0 and rp is not None:
                    rp = rp[:, None, :].expand(r.shape).contiguous()
                if mf == 0 and fp is not None:
                    fp = fp[:, None, :].expand(f.shape).contiguous()
                mr = mf = 1
            pairs.append((r, f, rp, fp, mr))
        sums, ema, lsums = ops.lsgan_d_prepare(pairs, self._state, self.decay, self.comm)
        return [ops.LsganDFn.apply(r, f, rp, fp, mode, sums[i], ema[i], lsums[i]) for i, (r, f, rp, fp, mode) in enumerate(pairs)]

    def discriminator_loss(self, real_output, fake_output, real_mask=None, fake_mask=None):
        return self.discriminator_losses([(real_output, fake_output, real_mask, fake_mask)])[0]

    def generator_loss(self, fake_output, fake_mask=None):
        f = _bhw(fake_output)
        pad, mode = _padded_map(fake_mask, f)
        return ops.LsganGFn.apply(f, pad, mode, self.comm)

    def generator_loss_deferred(self, fake_output, fake_mask, batch: "ops.SumsBatch"):
        """generator_loss in two halves: the sums go into ``batch`` now; call the returned function AFTER ``batch.reduce()``
        for the loss.  Lets every loss of a generator step share one all-reduce (mqgan_amd.train)."""
        f = _bhw(fake_output)
        pad, mode = _padded_map(fake_mask, f)
        prepared = ops.lsgan_g_sums(f, pad, mode, batch)
        return lambda: ops.LsganGFn.apply(f, pad, mode, None, prepared)


class MaskedMelLoss(nn.Module):
    def __init__(self, loss_type: str = "charbonnier", group_size: int = 1, eps: float = 1e-6, comm: Comm = None):
        super().__init__()
        assert loss_type in {"charbonnier", "mse"}
        if loss_type != "mse":
            raise NotImplementedError("only loss_type='mse' is on the hot path (train.py:217-218)")
        self.loss_type, self.group_size, self.eps, self.comm = loss_type, group_size, eps, comm

    def forward(self, x, y, lengths):
        assert x.shape == y.shape, "x and y must have the same shape"
        assert x.shape[-1] % self.group_size == 0, "C (n_mels) must be divisible by group_size"
        return ops.MelLossFn.apply(x.contiguous(), y.contiguous(), lengths.to(torch.int32).contiguous(), self.group_size,
                                   self.group_size, 1.0, 0.0, self.comm)


def recon_loss(x, y, lengths, group_size=16, group_weight=0.25, comm: Comm = None):
    """loss_all + 0.25 * loss_group in ONE pass (train.py:419-425)."""
    return ops.MelLossFn.apply(x.contiguous(), y.contiguous(), lengths.to(torch.int32).contiguous(), 1, group_size, 1.0,
                               float(group_weight), comm)


def recon_loss_deferred(x, y, lengths, batch: "ops.SumsBatch", group_size=16, group_weight=0.25):
    """recon_loss in two halves (see LSGANLoss.generator_loss_deferred)."""
    x, y, lengths = x.contiguous(), y.contiguous(), lengths.to(torch.int32).contiguous()
    prepared = ops.mel_loss_sums(x, y, lengths, batch)
    return lambda: ops.MelLossFn.apply(x, y, lengths, 1, group_size, 1.0, float(group_weight), None, prepared)


def masked_mae_deferred(pred, target, mask, batch: "ops.SumsBatch", leaky_slope=None, n_bands: int = 1):
    """masked_mae (n_bands == 1) / masked_mae_bands in two halves (see LSGANLoss.generator_loss_deferred)."""
    pred, target = ops.to_f32(pred).contiguous(), ops.to_f32(target).contiguous()
    if n_bands > 1:
        if pred.shape[0] % n_bands != 0 or tuple(mask.shape) != (pred.shape[0] // n_bands, pred.shape[2]):
            raise ValueError(f"masked_mae_bands: pred {tuple(pred.shape)}, mask {tuple(mask.shape)}, bands {n_bands}")
        mask = mask.contiguous()
        prepared = ops.masked_mae_sums(pred, target, mask, batch, bands=n_bands)
        return lambda: ops.MaskedMaeBandsFn.apply(pred, target, mask, n_bands, None, leaky_slope, prepared)
    prepared = ops.masked_mae_sums(pred, target, mask, batch)
    return lambda: ops.MaskedMaeFn.apply(pred, target, mask, None, leaky_slope, prepared)


def masked_mae_bands(pred, target, mask, n_bands: int, comm: Comm = None, leaky_slope=None):
    """masked_mae of each of the ``n_bands`` row blocks of a multi-band feature map -> (n_bands,) losses (one autograd node)."""
    if pred.shape[0] % n_bands != 0 or tuple(mask.shape) != (pred.shape[0] // n_bands, pred.shape[2]):
        raise ValueError(f"masked_mae_bands: pred {tuple(pred.shape)}, mask {tuple(mask.shape)}, bands {n_bands}")
    return ops.MaskedMaeBandsFn.apply(ops.to_f32(pred).contiguous(), ops.to_f32(target).contiguous(), mask.contiguous(), n_bands, comm,
                                      leaky_slope)


def masked_mae(pred, target, mask, eps: float = 1e-8, comm: Comm = None, leaky_slope=None):
    """train.py:38-45 on channels-last features (B, H, W, C) with a uint8 padded mask (B, W).  ``leaky_slope``: pred is a
    feature map of a discriminator run with ``pregate=True`` (the LeakyReLU backward of its layer rides on this gradient)."""
    return ops.MaskedMaeFn.apply(ops.to_f32(pred).contiguous(), ops.to_f32(target).contiguous(), mask, comm, leaky_slope)
