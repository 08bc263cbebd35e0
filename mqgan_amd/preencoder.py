"""PreEncoder (generator), UNetRefiner and ConvBlock2D with the reference's public API, constructor
arguments and state_dict layout (reference preencoder.py:15-24, 86-202, 205-301, 304-599), computed
by the HIP kernels of libmqgan_hip.so.

Public tensors keep the reference's conventions: spectrograms are (B, T, mel) fp32, ``x_lengths``
is (B,) int, boolean masks are True = padded with shape (B, 1, T).  Internally everything is
channels-last and masks are uint8 (B, T).
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from .attentions import APTx, ResidualBlock1D
from .nn_base import ConvParams, conv2d, linear, pad_channels
from .quantizer import FSQ


def sequence_mask(max_length, x_lengths):
    """Bool (batch, max_length), True = padded (preencoder.py:15-24)."""
    return ops.sequence_mask(x_lengths, int(max_length)).bool()


def _to_u8_mask(x_mask, B, T, device):
    """(B, 1, T) / (B, T) bool (True = padded) or None -> uint8 (B, T)."""
    if x_mask is None:
        return torch.zeros((B, T), dtype=torch.uint8, device=device)
    m = x_mask.reshape(B, T)
    return m.to(torch.uint8).contiguous()


# ----------------------------------------------------------------------------- UNet refiner
class ConvBlock(nn.Module):
    """preencoder.py:86-102: mask -> conv3x3 -> APTx -> drop -> conv3x3 -> APTx -> drop (+ x) -> mask."""

    def __init__(self, c_in, c_out, dropout=0.1):
        super().__init__()
        self.conv1 = ConvParams(c_out, c_in, (3, 3), "wn")
        self.conv2 = ConvParams(c_out, c_out, (3, 3), "wn")
        self.act = APTx()
        self.p_drop = dropout
        self.match = c_in == c_out

    def forward(self, x, mask, input_is_masked=False, need_dx=True, cin_segments=None):
        """x (N, T, F, C); mask uint8 (N, T).  With bf16 storage the block's tensors may carry zero channels beyond the layer's
        own (nn_base.pad_channels): ``cin_segments`` = (real, stored) channels per part of x, the output has pad_channels(c_out)."""
        Fq = x.shape[2]
        dmap = (Fq, 1, 1)
        b, g = self.act.params()
        x = ops.to_storage(x)                                      # bf16 tensors between the convs when storage is on
        if not input_is_masked:
            x = ops.pointwise(x, mask=mask, mask_post=True, dmap=dmap)
        cp = pad_channels(self.conv1.cout)
        y = ops.to_storage(conv2d(x, self.conv1, need_dx=need_dx, cin_segments=cin_segments, cout_pad=cp))
        y = ops.pointwise(y, act=ops.ACT_APTX, beta=b, gamma=g, p_drop=self.p_drop, training=self.training)
        y = ops.to_storage(conv2d(y, self.conv2, cin_segments=[(self.conv2.cin, cp)], cout_pad=cp))
        return ops.pointwise(y, act=ops.ACT_APTX, beta=b, gamma=g, p_drop=self.p_drop, training=self.training,
                             r_post=x if self.match else None, mask=mask, mask_post=True, dmap=dmap)


class DownBlock(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = ConvBlock(c_in, c_out)

    def forward(self, x, mask):
        m = ops.mask_pool(mask, 2, ceil_mode=False)                # padding stays padded (:63-65)
        fused = ops.mask_fusable(x.shape[-1])                      # the block's entry masked_fill rides on the pooling pass
        x = ops.AvgPoolT2Fn.apply(x, m if fused else None)         # time / 2 (preencoder.py:108)
        return self.conv(x, m, input_is_masked=fused, cin_segments=[(self.conv.conv1.cin, x.shape[-1])]), m


class UpBlock(nn.Module):
    def __init__(self, c_in, c_skip, c_out):
        super().__init__()
        self.conv = ConvBlock(c_in + c_skip, c_out)
        self.c_in, self.c_skip = c_in, c_skip

    def forward(self, x, skip, mask):
        m = ops.mask_upsample2(mask)                               # :68-70
        t_up = 2 * x.shape[1]
        dt = skip.shape[1] - t_up
        if dt > 0:                                                 # crop_to_match (:73-82); never hit when T % 2^depth == 0
            skip = skip[:, dt // 2: dt // 2 + t_up].contiguous()
        fused = ops.mask_fusable(x.shape[-1], skip.shape[-1])
        segs = [(self.c_in, x.shape[-1]), (self.c_skip, skip.shape[-1])]       # either part may be stored zero-padded
        x = ops.UpCatFn.apply(x, skip, m if fused else None)       # nearest x2 in time + channel concat (:124-129), masked
        return self.conv(x, m, input_is_masked=fused, cin_segments=segs), m


class UNetRefiner(nn.Module):
    """preencoder.py:134-202.  x (B, T, F, 1) channels-last (the reference's (B, 1, T, F)); x_mask uint8
    (B, T) -> residual (B, T, F_out)."""

    def __init__(self, in_channels, base_ch=128, depth=3, dropout=0.1, input_out_channels=(144, 128)):
        super().__init__()
        self.depth = depth
        chs = [base_ch * (2 ** i) for i in range(depth + 1)]
        self.pre = ConvBlock(in_channels, chs[0], dropout)
        self.downs = nn.ModuleList(DownBlock(chs[i], chs[i + 1]) for i in range(depth))
        self.mid = ConvBlock(chs[-1], chs[-1], dropout)
        self.ups = nn.ModuleList(UpBlock(chs[depth - i], chs[depth - i - 1], chs[depth - i - 1]) for i in range(depth))
        self.post = ConvParams(1, chs[0], (3, 3), "wn")
        self.reproj = ConvParams(input_out_channels[1], input_out_channels[0], (), bias=False)

    def forward(self, x, x_mask):
        with ops.act_storage(ops.ACT_BF16_REQUESTED):              # bf16 tensors inside, f32 in and out
            return self._forward(x, x_mask)

    def _forward(self, x, x_mask):
        B, T0, Fq, _ = x.shape
        mult = 1 << self.depth
        pad = (mult - T0 % mult) % mult                            # pad_to_pow2_4d (:29-47)
        if pad:
            xp = torch.zeros((B, T0 + pad, Fq, x.shape[3]), dtype=x.dtype, device=x.device)
            xp[:, :T0] = x
            mp = torch.ones((B, T0 + pad), dtype=torch.uint8, device=x.device)
            mp[:, :T0] = x_mask
            x, mask = xp, mp
        else:
            mask = x_mask
        skips = []
        x = self.pre(x, mask, need_dx=False)                       # the refiner input is detached (:413-415)
        cur = mask
        for down in self.downs:
            skips.append(x)
            x, cur = down(x, cur)
        x = self.mid(x, cur, input_is_masked=True, cin_segments=[(self.mid.conv1.cin, x.shape[-1])])
        for up in self.ups:
            x, cur = up(x, skips.pop(), cur)
        # x is already masked with `cur` by the last block, so apply_mask(x, cur_mask) (:191) is a no-op
        out = ops.to_f32(conv2d(x, self.post, cin_segments=[(self.post.cin, x.shape[-1])])).view(B, T0 + pad, Fq)
        if pad:
            out = out[:, :T0].contiguous()
        out = ops.pointwise(out, mask=x_mask, mask_post=True)      # (:197-198)
        return linear(out, self.reproj)


# ----------------------------------------------------------------------------- ConvBlock2D
class ConvBlock2D(nn.Module):
    """preencoder.py:205-301 in the only configuration the PreEncoder builds (:322, :351):
    in_channels=1, depthwise=True, kernel 5, act="aptx", weight norm, dropout 0.1 - as ONE fused kernel."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, dilation=1, *, depthwise=False,
                 use_weight_norm=True, act="relu", dropout=0.1, bias=True):
        super().__init__()
        if not (in_channels == 1 and depthwise and kernel_size == 5 and stride == 1 and dilation == 1 and use_weight_norm
                and act.lower() == "aptx" and bias):
            raise NotImplementedError("ConvBlock2D: only the PreEncoder configuration (1->C0, depthwise 5x5, aptx) is fused")
        self.dw = ConvParams(1, 1, (5, 5), "wn")
        self.pw = ConvParams(out_channels, 1, (1, 1), "wn")
        self.activation = APTx()
        self.p_drop = dropout
        self.depthwise = True
        self.conv_out = ConvParams(1, out_channels, (1, 1))

    def forward(self, x, x_mask):
        """x (B, T, C0) channels-last, x_mask uint8 (B, T) -> (B, T, C0)."""
        C0 = self.pw.cout
        wdw = self.dw.effective(need_crsk=False)[0].view(25)
        w1 = self.pw.effective(need_crsk=False)[0].view(C0)
        p = self.p_drop if self.training else 0.0
        seed = ops.next_seed() if p > 0 else 0
        return ops.ConvBlock2DFn.apply(x, x_mask, wdw, self.dw.bias, w1, self.pw.bias, self.conv_out.weight.view(C0),
                                       self.conv_out.bias, p, seed)


# ----------------------------------------------------------------------------- PreEncoder
class PreEncoder(nn.Module):
    def __init__(self, mel_channels, channels, kernel_sizes, fsq_levels=[8, 8, 5, 5, 5], dropout=0.1,
                 refiner_base_channels=128, refiner_depth=3, refiner_hidden_proj_divisor=8):
        super().__init__()
        self.proj = ConvParams(channels[0], mel_channels, ())
        self.pre = ConvBlock2D(1, channels[0], kernel_size=5, depthwise=True, act="aptx")
        self.quantizer_dim = len(fsq_levels)
        self.encoder_blocks = nn.ModuleList([
            ResidualBlock1D(channels[i], channels[i + 1], kernel_size=kernel_sizes[i], dropout=dropout, act="taptx",
                            norm="weight") for i in range(len(channels) - 1)])
        latent_dim = channels[-1]
        self.q_in_proj = ConvParams(self.quantizer_dim, latent_dim, ())
        self.quantizer = FSQ(levels=fsq_levels)
        self.q_out_proj = ConvParams(latent_dim, self.quantizer_dim, ())
        self.codebook_size = 1
        for level in fsq_levels:
            self.codebook_size *= level
        self.bos_token_id = self.codebook_size + 1
        self.eos_token_id = self.codebook_size + 2
        rev_channels, rev_kernel_sizes = list(reversed(channels)), list(reversed(kernel_sizes))
        self.decoder_blocks = nn.ModuleList([
            ResidualBlock1D(rev_channels[i], rev_channels[i + 1], kernel_size=rev_kernel_sizes[i], dropout=dropout,
                            act="taptx", causal=True, norm="weight") for i in range(len(rev_channels) - 1)])
        self.post = ConvBlock2D(1, channels[0], kernel_size=5, depthwise=True, act="aptx")
        self.out_proj = ConvParams(mel_channels, channels[0], ())
        self.refiner_hidden_channels = mel_channels // refiner_hidden_proj_divisor
        self.hidden_proj = ConvParams(self.refiner_hidden_channels, channels[0], ())
        self.refiner = UNetRefiner(in_channels=1, base_ch=refiner_base_channels, depth=refiner_depth,
                                   input_out_channels=[mel_channels + self.refiner_hidden_channels, mel_channels])
        self.mel_channels = mel_channels

    # -- halves ---------------------------------------------------------------------------------
    def _encode_latent(self, x, mask):
        h = linear(x.contiguous(), self.proj, need_dx=x.requires_grad)          # (B, T, C0)
        h = self.pre(h, mask)
        for block in self.encoder_blocks:
            h = block(h, x_mask=mask)
        return linear(h, self.q_in_proj)                                        # (B, T, d)

    def _decode_codes(self, codes, mask, return_hidden=False):
        h = linear(codes, self.q_out_proj)
        for block in self.decoder_blocks:
            h = block(h, x_mask=mask)
        dec = h
        x_recon = linear(self.post(dec, mask), self.out_proj)
        with torch.no_grad():                                                   # consumer is detached (preencoder.py:413-415)
            hid = linear(dec.detach(), self.hidden_proj, need_dx=False)
            rin = ops.Concat2Fn.apply(x_recon.detach(), hid)                    # (B, T, mel + hid)
        residual = self.refiner(rin.unsqueeze(-1), mask)
        x_post = ops.add(x_recon, residual)
        if return_hidden:
            return x_recon, x_post, dec
        return x_recon, x_post

    # -- reference API --------------------------------------------------------------------------
    def forward(self, x, x_lengths):
        """(B, T, mel), (B,) -> (x_recon, x_post), each (B, T, mel).  preencoder.py:363-418."""
        B, T, _ = x.shape
        mask = ops.sequence_mask(x_lengths.to(x.device), T)
        z = self._encode_latent(x, mask)
        codes, _ = self.quantizer(z)
        return self._decode_codes(codes, mask)

    def encode(self, x, x_mask=None):
        """(B, T, mel), bool (B, 1, T) True = padded or None -> int64 indices (B, T).  :420-451."""
        B, T, _ = x.shape
        mask = _to_u8_mask(x_mask, B, T, x.device)
        _, indices = self.quantizer(self._encode_latent(x, mask))
        return indices.long()

    def decode(self, indices, x_mask=None, return_hidden=False):
        """indices (B, T) -> x_post (B, T, mel) [, last_hid (B, C0, T)].  :453-504."""
        B, T = indices.shape
        mask = _to_u8_mask(x_mask, B, T, indices.device)
        codes = self.quantizer.indices_to_codes(indices)
        out = self._decode_codes(codes, mask, return_hidden=return_hidden)
        if return_hidden:
            return out[1], out[2].permute(0, 2, 1)
        return out[1]


def strip_weight_norm(module):
    """preencoder.py:507-514: fold the LEGACY weight-norm convs (decoder) into plain weights."""
    for m in module.modules():
        if isinstance(m, ConvParams):
            m.fold_weight_norm_()


def get_pre_encoder(model_path: str, device, channels=[384, 512, 768], kernel_sizes=[7, 5, 3], mel_channels=88,
                    fsq_levels=[8, 5, 5, 5], refiner_base_channels=128, refiner_depth=3, refiner_hidden_proj_divisor=8,
                    inference=False):
    """Load a PreEncoder from a training checkpoint (preencoder.py:517-599): eval mode, dropout 0,
    strict key match; raises FileNotFoundError / KeyError / RuntimeError like the reference."""
    if not os.path.isfile(model_path):
        raise FileNotFoundError(f"Checkpoint file not found: {model_path}")
    checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
    try:
        model = PreEncoder(mel_channels=mel_channels, channels=channels, kernel_sizes=kernel_sizes, dropout=0.0,
                           fsq_levels=fsq_levels, refiner_base_channels=refiner_base_channels, refiner_depth=refiner_depth,
                           refiner_hidden_proj_divisor=refiner_hidden_proj_divisor)
    except Exception as e:
        raise RuntimeError(f"Failed to instantiate model with loaded config: {e}")
    if "model_state_dict" not in checkpoint:
        raise KeyError("Checkpoint missing 'model_state_dict' key containing weights.")
    clean = OrderedDict((k[7:] if k.startswith("module.") else k, v) for k, v in checkpoint["model_state_dict"].items())
    model.load_state_dict(clean, strict=True)
    model.to(device)
    model.eval()
    if inference:
        strip_weight_norm(model)
    return model
