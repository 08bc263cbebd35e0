"""APTx, CBAM1D and ResidualBlock1D with the reference's names, constructor arguments and
state_dict keys (reference attentions.py:9-35, 195-419, 476-551), computed by the HIP kernels.

Interface difference (internal only): feature maps are channels-last (B, T, C) and the padding
mask is a uint8 (B, T) tensor, 1 = padded.  PreEncoder converts at its boundary.
"""
import torch
import torch.nn as nn

from . import ops
from .nn_base import ConvParams, conv1d, linear


class APTx(nn.Module):
    """(alpha + tanh(beta x)) * gamma * x with alpha = 1 (attentions.py:17-35)."""

    def __init__(self, alpha=1, beta=1, gamma=0.5, trainable=False):
        super().__init__()
        if alpha != 1:
            raise NotImplementedError("the HIP APTx kernel implements alpha = 1 (the only value the reference uses)")
        self.alpha = alpha
        self.trainable = trainable
        if trainable:
            self.beta = nn.Parameter(torch.tensor(float(beta), dtype=torch.float32))
            self.gamma = nn.Parameter(torch.tensor(float(gamma), dtype=torch.float32))
        else:
            self.beta_value, self.gamma_value = float(beta), float(gamma)
            self.register_buffer("_bg", torch.tensor([float(beta), float(gamma)], dtype=torch.float32), persistent=False)

    def params(self):
        if self.trainable:
            return self.beta, self.gamma
        return self._bg[0:1], self._bg[1:2]

    def forward(self, x):
        b, g = self.params()
        return ops.pointwise(x, act=ops.ACT_APTX, beta=b, gamma=g)


class _MLP(nn.Module):
    """nn.Sequential(Linear, ReLU, Linear) with keys ``0.*`` / ``2.*`` (attentions.py:211-215)."""

    def __init__(self, c, r):
        super().__init__()
        self.add_module("0", ConvParams(c // r, c, ()))
        self.add_module("2", ConvParams(c, c // r, ()))

    def forward(self, x):
        h = ops.ReluFn.apply(linear(x, getattr(self, "0")))
        return linear(h, getattr(self, "2"))


class CAM1D(nn.Module):
    def __init__(self, channels: int, reduction_ratio: int, causal: bool = False):
        super().__init__()
        if causal:
            raise NotImplementedError("causal CBAM is never instantiated by the reference's PreEncoder")
        self.channels, self.r, self.causal = channels, reduction_ratio, causal
        self.mlp = _MLP(channels, reduction_ratio)

    def gate(self, x, mask):
        """sigmoid(mlp(max over all t) + mlp(mean over valid t)) -> (B, C).  The max ignores the mask
        (reference quirk: attentions.py:94 discards the result of the non-in-place masked_fill_)."""
        mx, av = ops.CamPoolFn.apply(x, mask)
        both = self.mlp(torch.cat([mx, av], dim=0))          # one GEMM for both pooled vectors
        B = x.shape[0]
        return ops.SigmoidAddFn.apply(both[:B].contiguous(), both[B:].contiguous())


class SAM1D(nn.Module):
    def __init__(self, kernel_size: int = 7, bias: bool = False, causal: bool = False):
        super().__init__()
        if causal or bias:
            raise NotImplementedError("only the non-causal, bias-free SAM of the reference is implemented")
        self.conv = ConvParams(1, 2, (kernel_size,), bias=False)


class CBAM1D(nn.Module):
    """attentions.py:367-419.  Output = SAM(CAM(x)) + x, not re-masked (reference quirk)."""

    def __init__(self, channels: int, reduction_ratio: int = 8, causal: bool = False, sam_kernel_size: int = 7):
        super().__init__()
        self.causal = causal
        self.channel_attention = CAM1D(channels, reduction_ratio, causal)
        self.spatial_attention = SAM1D(sam_kernel_size, bias=False, causal=causal)

    def forward(self, x, mask):
        gate = self.channel_attention.gate(x, mask)
        return ops.CbamApplyFn.apply(x, gate, self.spatial_attention.conv.weight.view(2, -1))


class ResidualBlock1D(nn.Module):
    """attentions.py:476-551 with norm="weight" (the only normalisation the PreEncoder uses)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, dilation=1, dropout=0.3, act="relu", causal=False,
                 norm="layer"):
        super().__init__()
        if norm != "weight":
            raise NotImplementedError("ResidualBlock1D: only norm='weight' is on the hot path")
        if act not in ("taptx", "aptx"):
            raise NotImplementedError(f"ResidualBlock1D: activation {act!r} is not on the hot path")
        self.causal, self.dilation, self.p_drop = causal, dilation, dropout
        flavour = "wn_legacy" if causal else "wn"
        self.conv1 = ConvParams(out_channels, in_channels, (kernel_size,), flavour)
        self.conv2 = ConvParams(out_channels, out_channels, (kernel_size,), flavour)
        self.cbam = None if causal else CBAM1D(out_channels, causal=False)
        self.relu = APTx(trainable=(act == "taptx"))
        self.residual = ConvParams(out_channels, in_channels, (1,)) if in_channels != out_channels else None

    def forward(self, x, x_mask=None):
        """x (B, T, Cin), x_mask uint8 (B, T) 1 = padded -> (B, T, Cout)."""
        beta, gamma = self.relu.params()
        masked = x_mask is not None
        res = conv1d(x, self.residual, causal=False) if self.residual is not None else x
        out = conv1d(x, self.conv1, self.causal, self.dilation)
        out = ops.pointwise(out, act=ops.ACT_APTX, mask=x_mask, mask_pre=masked, beta=beta, gamma=gamma)
        out = conv1d(out, self.conv2, self.causal, self.dilation)
        if self.cbam is not None:
            if x_mask is None:
                x_mask = torch.zeros(x.shape[:2], dtype=torch.uint8, device=x.device)
            out = self.cbam(out, x_mask)
        return ops.pointwise(out, act=ops.ACT_APTX, r_pre=res, mask=x_mask, mask_pre=masked, beta=beta, gamma=gamma,
                             p_drop=self.p_drop, training=self.training)
