"""Patch and multi-bin discriminators with the reference's API and state_dict keys (reference
discriminators.py:10-67, 70-242, 245-311) on the HIP kernels.

Both run through one code path: the (H = mel, W = time) single-channel image of every band is
stacked along the batch and each conv is ONE batch-grouped implicit-GEMM launch (groups = number of
bands; 1 for the patch discriminator), instead of eight small independent networks.
"""
from typing import List, Tuple, Union

import torch
import torch.nn as nn

from . import ops
from .nn_base import ConvParams, share_effective_weights


class ChannelSELayerMasked(nn.Module):
    """discriminators.py:10-67 (keys fc1.*, fc2.*)."""

    def __init__(self, num_channels: int, reduction_ratio: int = 2):
        super().__init__()
        reduced = max(1, num_channels // reduction_ratio)
        self.fc1 = ConvParams(reduced, num_channels, ())
        self.fc2 = ConvParams(num_channels, reduced, ())


def _layer_strides(stride, n_convs):
    if isinstance(stride, int):
        ls = [(1, stride)] * n_convs
    elif isinstance(stride, tuple) and len(stride) == 2:
        ls = [stride] * n_convs
    else:
        assert len(stride) == n_convs, "stride list must match kernel_sizes"
        ls = [tuple(s) for s in stride]
    return ls[:-1] + [(1, 1)]                      # the logits conv is always stride 1 (discriminators.py:165)


class MelSpectrogramPatchDiscriminator2D(nn.Module):
    def __init__(self, mel_channels: int, hidden_channels: list = (64, 128, 256, 512), kernel_sizes: list = (7, 5, 5, 3, 3),
                 stride: Union[int, Tuple[int, int], List[Tuple[int, int]]] = (2, 2), lengthwise_only=False):
        super().__init__()
        assert len(kernel_sizes) == len(hidden_channels) + 1, "kernel_sizes must be hidden_channels len + 1"
        if lengthwise_only:
            raise NotImplementedError("lengthwise_only discriminators are not used by train.py")
        self.mel_channels = mel_channels
        n = len(kernel_sizes)
        self.ret_features_map = [False, False] + [True] * (n - 3) + [False]
        self.strides = _layer_strides(stride, n)
        self.convs = nn.ModuleList()
        in_ch = 1
        for out_ch, k in zip(list(hidden_channels) + [1], kernel_sizes):
            k1, k2 = (k, k) if not isinstance(k, (tuple, list)) else k
            self.convs.append(ConvParams(out_ch, in_ch, (k1, k2), "sn", init="normal02"))
            in_ch = out_ch
        self.se_block = ChannelSELayerMasked(hidden_channels[-1], 8)

    # reference API: (B, T, F), (B,) -> logits (B,1,H,W), patch_mask bool True = valid [, features]
    def forward(self, x, x_lengths, return_features: bool = False):
        logits, mask, feats = run_discriminators([self], x, x_lengths, return_features)
        B, H, W, _ = logits.shape
        valid = (~mask.bool())[:, None, None, :].expand(B, 1, H, W)
        if return_features:
            fl = [(f.permute(0, 3, 1, 2), m.bool()[:, None, None, :].expand(f.shape[0], 1, f.shape[1], f.shape[2]))
                  for f, m in feats]
            return logits.permute(0, 3, 1, 2), valid, fl
        return logits.permute(0, 3, 1, 2), valid


class MultiBinDiscriminator(nn.Module):
    def __init__(self, mel_channels: int, n_bins: int = 4, hidden_channels: list = (64, 128, 256, 512),
                 kernel_sizes: list = (7, 5, 5, 3, 3), n_no_strides: int = 2):
        super().__init__()
        assert mel_channels % n_bins == 0, "mel_channels must divide n_bins"
        self.n_bins = n_bins
        strides = [(1, 1) if i < n_no_strides else (1, 2) for i in range(len(kernel_sizes))]
        self.discriminators = nn.ModuleList([
            MelSpectrogramPatchDiscriminator2D(mel_channels // n_bins, hidden_channels=list(hidden_channels),
                                               kernel_sizes=[(3, ks) for ks in kernel_sizes], stride=strides)
            for _ in range(n_bins)])

    def forward(self, x, x_lengths, return_features: bool = False):
        logits, mask, feats = run_discriminators(list(self.discriminators), x, x_lengths, return_features)
        G = self.n_bins
        B = logits.shape[0] // G
        H, W = logits.shape[1], logits.shape[2]
        valid = (~mask[:B].bool())[:, None, None, :].expand(B, 1, H, W)
        outs = [logits[j * B:(j + 1) * B].permute(0, 3, 1, 2) for j in range(G)]
        masks = [valid for _ in range(G)]
        if return_features:
            fl = [[(f[j * B:(j + 1) * B].permute(0, 3, 1, 2),
                    m[:B].bool()[:, None, None, :].expand(B, 1, f.shape[1], f.shape[2])) for f, m in feats] for j in range(G)]
            return outs, masks, fl
        return outs, masks


def _stacked(bands, get, need_crsk=True, power_iteration=False, detach=False):
    """Effective weights of the same layer of every band, stacked along the group axis."""
    shared = share_effective_weights.active
    key = ("stacked", id(get(bands[0])), len(bands), need_crsk, detach)
    if shared is not None and not power_iteration and key in shared:
        return shared[key]
    out = _stack_now(bands, get, need_crsk, power_iteration, detach)
    if shared is not None and not power_iteration:
        shared[key] = out
    return out


def _stack_now(bands, get, need_crsk, power_iteration, detach):
    mods = [get(b) for b in bands]
    if (1 < len(mods) <= ops.SN_MAX_BANDS and not power_iteration and all(m.flavour == "sn" for m in mods)
            and not any(getattr(m, "_frozen", False) for m in mods) and mods[0].weight_orig.is_cuda
            and all(tuple(m.weight_orig.shape) == tuple(mods[0].weight_orig.shape) for m in mods)):
        # eval-mode band discriminators: spectral norm of the same layer of all bands in two launches, written stacked
        w_krsc, w_crsk = ops.spectral_prep_bands([m.weight_orig for m in mods], [m.weight_u for m in mods],
                                                 [m.weight_v for m in mods], need_crsk)
        bias = torch.cat([m.bias for m in mods], dim=0)
        if detach:
            w_krsc, bias = w_krsc.detach(), bias.detach()
        return w_krsc, w_crsk, bias
    ws = [m.effective(need_crsk=need_crsk, power_iteration=power_iteration) for m in mods]
    if len(ws) == 1:
        w_krsc, w_crsk = ws[0]
        bias = mods[0].bias
    else:
        w_krsc = torch.cat([w[0] for w in ws], dim=0)
        w_crsk = torch.cat([w[1] for w in ws], dim=0) if need_crsk else None
        bias = torch.cat([m.bias for m in mods], dim=0)
    if detach:
        w_krsc, bias = w_krsc.detach(), bias.detach()
    return w_krsc, w_crsk, bias


def run_discriminators(bands: List[MelSpectrogramPatchDiscriminator2D], x, x_lengths, return_features=False,
                       need_input_grad=None, detach_params=False, pregate=False):
    """Shared forward.  x (B, T, F) -> logits (G*B, H, W, 1) channels-last, padded mask uint8 (G*B, W) and
    [(feature (G*B, H, W, C), padded mask (G*B, W))].  G = len(bands); band j owns rows [jB, (j+1)B).
    ``detach_params``: treat the weights as constants (generator step: no discriminator weight gradients).
    ``pregate`` (the trainer's mode): the LeakyReLU / mask backward of a hidden layer rides on the store of the next conv's
    input-gradient kernel (ops.conv_leaky); features then come back as (feature, mask, leaky_slope or None) and whoever
    differentiates through a feature with a slope has to gate the gradient (losses.masked_mae(leaky_slope=...))."""
    with ops.act_storage(ops.ACT_BF16_REQUESTED):                  # bf16 feature maps inside, f32 logits / features out
        return _run_discriminators(bands, x, x_lengths, return_features, need_input_grad, detach_params, pregate)


# Parity-test hook: a list here receives, per _run_discriminators call, the list over layers of (activation > 0) as host bool
# tensors (G*B, H, W, C) - which branch every LeakyReLU took.  A float64 / oracle twin of the step can then take the same
# branches: two f32 algorithms disagree about the sign of a pre-activation within rounding of zero, and one such element changes
# a whole receptive field of the gradients; which branch it takes is not a property of the convolutions under test.
GATE_CAPTURE = None


def _run_discriminators(bands, x, x_lengths, return_features, need_input_grad, detach_params, pregate=False):
    G = len(bands)
    captured = [] if GATE_CAPTURE is not None else None
    d0 = bands[0]
    B, T, Fq = x.shape
    training = d0.training
    need_dx = x.requires_grad if need_input_grad is None else need_input_grad
    mask = ops.sequence_mask(x_lengths.to(x.device), T)                       # discriminators.py:184-193
    if G > 1:
        mask = mask.repeat(G, 1)
    out = ops.BandSplitFn.apply(x.contiguous(), G)                            # (G*B, F/G, T, 1)  (:211, :298)
    feats = []
    n = len(d0.convs)
    gate = None                                                               # (mask, dmap, slope) of a pregated `out`
    for i in range(n):
        N, H, W, Cin = out.shape
        if i == n - 1:                                                        # masked SE before the logits conv (:216-217)
            out = ops.to_f32(out)
            out, sq = ops.SeSqueezeSkipFn.apply(out, mask)                    # (G*B, C); `out` itself, its gradient summed in place
            w1k, w1c, b1 = _stacked(bands, lambda d: d.se_block.fc1, detach=detach_params)
            w2k, w2c, b2 = _stacked(bands, lambda d: d.se_block.fc2, detach=detach_params)
            r = b1.numel() // G
            h = ops.conv(sq.view(N, 1, 1, Cin), w1k, w1c, b1, ops.conv_desc(N, 1, 1, Cin, r, 1, 1, groups=G))
            h = ops.ReluFn.apply(h)
            e = ops.conv(h, w2k, w2c, b2, ops.conv_desc(N, 1, 1, r, Cin, 1, 1, groups=G))
            e = ops.SigmoidAddFn.apply(e.view(N, Cin), None)
            out = ops.SeScaleFn.apply(out, e)
        conv_i = d0.convs[i]
        kh, kw = conv_i.ksize
        sh, sw = d0.strides[i]
        first = (i == 0)
        wk, wc, bias = _stacked(bands, lambda d: d.convs[i], need_crsk=(need_dx or not first), power_iteration=training,
                                detach=detach_params)
        desc = ops.conv_desc(N, H, W, Cin, conv_i.cout, kh, kw, stride=(sh, sw), pad=((kh - 1) // 2, (kw - 1) // 2), groups=G)
        if sw > 1:                                                            # ceil-mode max-pool of the mask (:223-229)
            mask = ops.mask_pool(mask, sw, ceil_mode=True)
        # conv -> LeakyReLU(0.2) -> masked_fill (:219, :232), fused into the conv's store where the kernel allows it
        # (the layer in front of the squeeze-excite block and the logits conv hand f32 tensors on)
        dmap = (desc.P * desc.Q, desc.Q, desc.Q)
        if pregate:
            # layers whose output feeds the next conv (not the squeeze-excite block / the logits) may hand their activation
            # backward to that conv's dgrad store
            out, honoured = ops.conv_leaky(out, wk, wc, bias, desc, mask, dmap, out_f32=(i >= n - 2), in_gate=gate,
                                           pregate=(i < n - 2), chain=True)
            gate = (mask, dmap, 0.2) if honoured else None
            if return_features and d0.ret_features_map[i]:
                feats.append((ops.to_f32(out), mask, 0.2 if honoured else None))
        else:
            out = ops.conv_leaky(out, wk, wc, bias, desc, mask, dmap, out_f32=(i >= n - 2))
            if return_features and d0.ret_features_map[i]:
                feats.append((ops.to_f32(out), mask))
        if captured is not None:
            captured.append((ops.to_f32(out).detach() > 0).cpu())
    if captured is not None:
        GATE_CAPTURE.append(captured)
    return ops.to_f32(out), mask, feats
