"""LSGANLoss (+ LeCam) and MaskedMelLoss with the reference's API (reference losses.py:5-116,
121-182) and masked_mae (train.py:38-45), on the HIP kernels.  No host synchronisation: the
reference's ``valid.item() > 0`` branch (losses.py:29) is taken on the device.
"""
import torch
import torch.nn as nn

from . import ops


def _padded_bw(mask, logits):
    """Loss mask of the reference (bool, True = VALID, shape of the logits (B, 1, H, W)) -> uint8 padded
    mask (B, W).  Discriminator masks are constant along H, which is all the kernels need."""
    if mask is None:
        return torch.zeros((logits.shape[0], logits.shape[-1]), dtype=torch.uint8, device=logits.device)
    if mask.dtype == torch.uint8 and mask.dim() == 2:
        return mask
    m = mask.reshape(mask.shape[0], -1, mask.shape[-1])[:, 0, :]
    return (~m.bool()).to(torch.uint8).contiguous()


def _bhw(logits):
    """(B, 1, H, W) or channels-last (B, H, W, 1) single-channel logits -> contiguous (B, H, W)."""
    if logits.dim() != 4:
        raise ValueError(f"expected 4-D single-channel logits, got {tuple(logits.shape)}")
    if logits.shape[1] == 1 and logits.shape[3] != 1:
        return logits.reshape(logits.shape[0], logits.shape[2], logits.shape[3])
    return logits.reshape(logits.shape[0], logits.shape[1], logits.shape[2])


class LSGANLoss(nn.Module):
    def __init__(self, real_label=1.0, fake_label=0.0, decay=0.99, use_lecam=True):
        super().__init__()
        if real_label != 1.0 or fake_label != 0.0 or not use_lecam:
            raise NotImplementedError("the HIP LSGAN kernel implements labels (1, 0) with LeCam, as train.py uses it")
        self.real_label, self.fake_label, self.decay, self.use_lecam = real_label, fake_label, decay, use_lecam
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

    def discriminator_loss(self, real_output, fake_output, real_mask=None, fake_mask=None):
        padded = _padded_bw(real_mask, real_output)
        return ops.LsganDFn.apply(_bhw(real_output).contiguous(), _bhw(fake_output).contiguous(), padded, self._state,
                                  float(self.decay))

    def generator_loss(self, fake_output, fake_mask=None):
        return ops.LsganGFn.apply(_bhw(fake_output).contiguous(), _padded_bw(fake_mask, fake_output))


class MaskedMelLoss(nn.Module):
    def __init__(self, loss_type: str = "charbonnier", group_size: int = 1, eps: float = 1e-6):
        super().__init__()
        assert loss_type in {"charbonnier", "mse"}
        if loss_type != "mse":
            raise NotImplementedError("only loss_type='mse' is on the hot path (train.py:217-218)")
        self.loss_type, self.group_size, self.eps = loss_type, group_size, eps

    def forward(self, x, y, lengths):
        assert x.shape == y.shape, "x and y must have the same shape"
        assert x.shape[-1] % self.group_size == 0, "C (n_mels) must be divisible by group_size"
        return ops.MelLossFn.apply(x.contiguous(), y.contiguous(), lengths.to(torch.int32).contiguous(), self.group_size,
                                   self.group_size, 1.0, 0.0)


def recon_loss(x, y, lengths, group_size=16, group_weight=0.25):
    """loss_all + 0.25 * loss_group in ONE pass (train.py:419-425)."""
    return ops.MelLossFn.apply(x.contiguous(), y.contiguous(), lengths.to(torch.int32).contiguous(), 1, group_size, 1.0,
                               float(group_weight))


def masked_mae(pred, target, mask, eps: float = 1e-8):
    """train.py:38-45 on channels-last features (B, H, W, C) with a uint8 padded mask (B, W)."""
    return ops.MaskedMaeFn.apply(pred.contiguous(), target.contiguous(), mask)
