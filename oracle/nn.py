"""Plain-PyTorch CPU restatement of the modules on the MQGAN hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional style: every function
takes ``P``, a dict ``{state_dict key: tensor}`` with the reference's key names,
plus a key prefix, so the golden fixtures' weights plug straight in.  Tensors
keep the reference's layouts ((B, C, T) for the 1-D stacks, (B, C, H, W) for the
2-D ones).  Autograd is torch's own: leaf tensors in ``P`` with
``requires_grad=True`` collect the gradients the tests compare.

Each function cites the reference lines it follows (paths are relative to the
reference checkout).  Quirks of the reference are reproduced on purpose and are
flagged QUIRK.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------- masks
def sequence_mask(max_length: int, lengths: torch.Tensor) -> torch.Tensor:
    """preencoder.py:15-24 - (B, max_length) bool, True = padded."""
    return torch.arange(max_length, device=lengths.device)[None, :] >= lengths[:, None]


# --------------------------------------------------------------------------- weights
def wn_weight(P: Params, prefix: str) -> torch.Tensor:
    """Effective weight of a weight-normalised conv.

    New parametrisation API (preencoder.py:53,248; attentions.py:500-501): keys
    ``parametrizations.weight.original0`` (g) / ``original1`` (v).  Legacy API
    (attentions.py:469): keys ``weight_g`` / ``weight_v``.  Both: w = g * v / ||v||,
    norm over every dim but 0.  Plain convs just have ``weight``.
    """
    if prefix + "weight" in P:
        return P[prefix + "weight"]
    if prefix + "weight_g" in P:
        g, v = P[prefix + "weight_g"], P[prefix + "weight_v"]
    else:
        g = P[prefix + "parametrizations.weight.original0"]
        v = P[prefix + "parametrizations.weight.original1"]
    norm = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / norm)


def sn_weight(P: Params, prefix: str, training: bool, eps: float = 1e-12) -> torch.Tensor:
    """Legacy ``torch.nn.utils.spectral_norm`` (discriminators.py:139,162).

    One power iteration per *training-mode* forward, updating ``weight_u`` /
    ``weight_v`` in place under no_grad; none in eval mode.  sigma = u^T W v is
    differentiable wrt ``weight_orig`` with u, v held constant.
    """
    w = P[prefix + "weight_orig"]
    u, v = P[prefix + "weight_u"], P[prefix + "weight_v"]
    wm = w.flatten(1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


# --------------------------------------------------------------------------- activations
def aptx(x, beta=1.0, gamma=0.5, alpha=1.0):
    """attentions.py:34-35."""
    return (alpha + torch.tanh(beta * x)) * gamma * x


def dropout(x, p: float, training: bool):
    return F.dropout(x, p, training) if (training and p > 0) else x


# --------------------------------------------------------------------------- FSQ (row Q1/Q2)
def fsq_consts(levels: List[int]):
    lv = torch.tensor(levels, dtype=torch.int32)
    half_l = (lv - 1) * (1 + 1e-3) / 2                       # quantizer.py:111
    offset = torch.where(lv % 2 == 0, 0.5, 0.0)              # :112
    shift = (offset / half_l).atanh()                        # :113
    half_width = lv // 2                                     # :132
    basis = torch.cumprod(torch.tensor([1] + levels[:-1]), dim=0, dtype=torch.int32)  # :71
    return lv, half_l, offset, shift, half_width, basis


def fsq_forward(z: torch.Tensor, levels: List[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """quantizer.py:207-272 with dim == len(levels) (no projections), eval mode or
    noise_dropout == 0 (train-mode branch :147-160 is then value-identical).
    z (..., d) fp32 -> codes (..., d) fp32 with STE gradient, indices (...) int32."""
    lv, half_l, offset, shift, half_width, basis = fsq_consts(levels)
    z = z.float()
    b = (z + shift).tanh() * half_l - offset                 # :114
    q = b + (b.round() - b).detach()                         # :39-42 round_ste (half-to-even)
    codes = q / half_width                                   # :137
    idx = ((codes * half_width + half_width) * basis).sum(dim=-1).to(torch.int32)  # :164-166,177-181
    return codes, idx


def fsq_indices_to_codes(idx: torch.Tensor, levels: List[int]) -> torch.Tensor:
    """quantizer.py:172-175,183-205."""
    lv, _, _, _, half_width, basis = fsq_consts(levels)
    lvl_idx = (idx.unsqueeze(-1) // basis) % lv
    return (lvl_idx - half_width) / half_width


# --------------------------------------------------------------------------- ConvBlock2D (row P1)
def convblock2d(P: Params, prefix: str, x: torch.Tensor, mask: torch.Tensor, training=False, p_drop=0.1,
                return_hidden=False):
    """preencoder.py:277-301 as built at :322/:351 (depthwise 5x5, APTx, dropout 0.1).

    x (B, C0, T) treated as a 1-channel C0 x T image; mask (B, 1, T) True = padded.
    QUIRK: the output is not re-masked, so padded frames come out as conv_out.bias.
    """
    x4 = x.unsqueeze(1)
    m4 = mask.unsqueeze(1)
    w_dw = wn_weight(P, prefix + "dw.")
    k = w_dw.shape[-1]
    out = F.conv2d(x4, w_dw, P[prefix + "dw.bias"], padding=k // 2)
    out = out.masked_fill(m4.expand_as(out), 0.0)
    out = F.conv2d(out, wn_weight(P, prefix + "pw."), P[prefix + "pw.bias"])
    out = out.masked_fill(m4.expand_as(out), 0.0)
    hidden = aptx(out)
    out = dropout(hidden, p_drop, training)
    out = F.conv2d(out, P[prefix + "conv_out.weight"], P[prefix + "conv_out.bias"]).squeeze(1)
    return (out, hidden) if return_hidden else out


# --------------------------------------------------------------------------- CBAM (row P3)
def cbam1d(P: Params, prefix: str, x: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """attentions.py:393-419 (CAM1D :217-273, SAM1D :310-365), non-causal.

    QUIRK: the helper ``masked_fill_`` (attentions.py:52-79) is NOT in place and
    its result is dropped at every call site, so: the max-pool runs over padded
    frames too; SAM's pooled features, logits, map and output are not masked; the
    CBAM output is not masked.  Only ``masked_avg_pool1d`` (:98-132) really masks.
    """
    # channel attention
    mx = x.max(dim=-1).values                                                  # :92-96 (mask dropped)
    valid = (~mask).expand(-1, x.shape[1], -1)
    avg = x.masked_fill(~valid, 0.0).sum(-1) / valid.sum(-1).clamp(min=1.0)    # :109-132

    def mlp(t):
        h = F.relu(F.linear(t, P[prefix + "channel_attention.mlp.0.weight"], P[prefix + "channel_attention.mlp.0.bias"]))
        return F.linear(h, P[prefix + "channel_attention.mlp.2.weight"], P[prefix + "channel_attention.mlp.2.bias"])

    gate = torch.sigmoid(mlp(mx) + mlp(avg)).unsqueeze(-1)                     # :258-265
    xc = gate * x                                                              # :268
    # spatial attention
    feat = torch.cat((xc.max(dim=1, keepdim=True).values, xc.mean(dim=1, keepdim=True)), dim=1)  # :322-333
    w = P[prefix + "spatial_attention.conv.weight"]
    logits = F.conv1d(feat, w, None, padding=w.shape[-1] // 2)                 # :343
    xs = torch.sigmoid(logits) * xc                                            # :351,359
    return xs + x                                                              # :411


# --------------------------------------------------------------------------- ResidualBlock1D (rows P2, P4)
def resblock1d(P: Params, prefix: str, x: torch.Tensor, mask: Optional[torch.Tensor], causal: bool,
               training=False, p_drop=0.0) -> torch.Tensor:
    """attentions.py:525-551 with norm="weight", act="taptx" (trainable beta/gamma
    shared by both activations).  Non-causal blocks have CBAM, causal ones left-pad
    d*(k-1) (CausalConv1da, :471-474) and have none."""
    beta, gamma = P[prefix + "relu.beta"], P[prefix + "relu.gamma"]

    def conv(name, t):
        w = wn_weight(P, prefix + name + ".")
        k = w.shape[-1]
        if causal:
            return F.conv1d(F.pad(t, (k - 1, 0)), w, P[prefix + name + ".bias"])
        return F.conv1d(t, w, P[prefix + name + ".bias"], padding=k // 2)    # padding="same", odd k

    if prefix + "residual.weight" in P:
        res = F.conv1d(x, P[prefix + "residual.weight"], P[prefix + "residual.bias"])
    else:
        res = x
    out = conv("conv1", x)
    if mask is not None:
        out = out.masked_fill(mask, 0)
    out = aptx(out, beta, gamma)
    out = conv("conv2", out)
    if not causal:
        out = cbam1d(P, prefix + "cbam.", out, mask)
    out = out + res
    if mask is not None:
        out = out.masked_fill(mask, 0)
    out = aptx(out, beta, gamma)
    return dropout(out, p_drop, training)


# --------------------------------------------------------------------------- UNet refiner (row P5)
def _refiner_convblock(P, prefix, x, mask4, training, p_drop=0.1):
    """preencoder.py:86-102 (dropout is the ctor default 0.1 everywhere: the
    PreEncoder never forwards its own dropout to the refiner, :358-359)."""
    if mask4 is not None:
        x = x.masked_fill(mask4, 0.0)
    y = F.conv2d(x, wn_weight(P, prefix + "conv1."), P[prefix + "conv1.bias"], padding=1)
    y = dropout(aptx(y), p_drop, training)
    y = F.conv2d(y, wn_weight(P, prefix + "conv2."), P[prefix + "conv2.bias"], padding=1)
    y = dropout(aptx(y), p_drop, training)
    if y.shape[1] == x.shape[1]:
        y = y + x
    if mask4 is not None:
        y = y.masked_fill(mask4, 0.0)
    return y


def unet_refiner(P: Params, prefix: str, x: torch.Tensor, x_mask: torch.Tensor, depth: int, training=False):
    """preencoder.py:169-202.  x (B, 1, T, F); x_mask (B, T) True = padded -> (B, T, F_out)."""
    T0 = x.shape[2]
    mult = 1 << depth
    pad_len = (mult - T0 % mult) % mult                                       # :29-47
    x = F.pad(x, (0, 0, 0, pad_len))
    mask4 = F.pad(x_mask, (0, pad_len), value=True)[:, None, :, None]
    skips = []
    x = _refiner_convblock(P, prefix + "pre.", x, mask4, training)
    cur = mask4
    for i in range(depth):                                                    # :179-181, 105-114
        skips.append(x)
        x = F.avg_pool2d(x, kernel_size=(2, 1))
        cur = F.max_pool2d(cur.float(), kernel_size=(2, 1), stride=(2, 1)).bool()
        x = _refiner_convblock(P, prefix + f"downs.{i}.conv.", x, cur, training)
    x = _refiner_convblock(P, prefix + "mid.", x, cur, training)
    for i in range(depth):                                                    # :187-189, 117-130
        skip = skips.pop()
        x = F.interpolate(x, scale_factor=(2, 1), mode="nearest")
        cur = F.interpolate(cur.float(), scale_factor=(2, 1), mode="nearest").bool()
        dt = skip.shape[-2] - x.shape[-2]
        if dt > 0:
            skip = skip[..., dt // 2: dt // 2 + x.shape[-2], :]
        x = torch.cat([x, skip], dim=1)
        x = _refiner_convblock(P, prefix + f"ups.{i}.conv.", x, cur, training)
    x = x.masked_fill(cur, 0.0)
    out = F.conv2d(x, wn_weight(P, prefix + "post."), P[prefix + "post.bias"], padding=1).squeeze(1)
    out = out[:, :T0, :]
    out = out.masked_fill(x_mask.unsqueeze(-1), 0.0)
    return F.linear(out, P[prefix + "reproj.weight"])


# --------------------------------------------------------------------------- PreEncoder (row P6)
def _n_blocks(P: Params, stem: str) -> int:
    n = 0
    while f"{stem}.{n}.conv1.bias" in P:
        n += 1
    return n


def _refiner_depth(P: Params) -> int:
    n = 0
    while f"refiner.downs.{n}.conv.conv1.bias" in P:
        n += 1
    return n


def preencoder_encode_latent(P: Params, x: torch.Tensor, mask3: torch.Tensor, training=False, p_drop=0.0):
    """First half of forward/encode (preencoder.py:375-389): -> z (B, T, d)."""
    h = F.linear(x, P["proj.weight"], P["proj.bias"]).permute(0, 2, 1)
    h = convblock2d(P, "pre.", h, mask3, training)
    for i in range(_n_blocks(P, "encoder_blocks")):
        h = resblock1d(P, f"encoder_blocks.{i}.", h, mask3, causal=False, training=training, p_drop=p_drop)
    return F.linear(h.permute(0, 2, 1), P["q_in_proj.weight"], P["q_in_proj.bias"])


def preencoder_decode_codes(P: Params, codes: torch.Tensor, mask3: torch.Tensor, training=False, p_drop=0.0,
                            return_hidden=False):
    """Second half (preencoder.py:391-416 / 466-499): codes (B, T, d) -> (x_recon, x_post[, last_hid])."""
    h = F.linear(codes, P["q_out_proj.weight"], P["q_out_proj.bias"]).permute(0, 2, 1)
    for i in range(_n_blocks(P, "decoder_blocks")):
        h = resblock1d(P, f"decoder_blocks.{i}.", h, mask3, causal=True, training=training, p_drop=p_drop)
    dec = h
    rc = convblock2d(P, "post.", dec, mask3, training).permute(0, 2, 1)
    x_recon = F.linear(rc, P["out_proj.weight"], P["out_proj.bias"])
    hid = F.linear(dec.permute(0, 2, 1), P["hidden_proj.weight"], P["hidden_proj.bias"])
    rin = torch.cat([x_recon, hid], dim=2).unsqueeze(1).detach()              # :410-415 (detach!)
    residual = unet_refiner(P, "refiner.", rin, mask3.squeeze(1), _refiner_depth(P), training)
    x_post = x_recon + residual
    if return_hidden:
        return x_recon, x_post, dec
    return x_recon, x_post


def preencoder_forward(P: Params, x: torch.Tensor, lengths: torch.Tensor, levels: List[int], training=False,
                       p_drop=0.0):
    """preencoder.py:363-418 -> (x_recon, x_post)."""
    mask3 = sequence_mask(x.shape[1], lengths).unsqueeze(1)
    z = preencoder_encode_latent(P, x, mask3, training, p_drop)
    codes, _ = fsq_forward(z, levels)
    return preencoder_decode_codes(P, codes, mask3, training, p_drop)


def preencoder_encode(P: Params, x: torch.Tensor, mask3: Optional[torch.Tensor], levels: List[int]):
    """preencoder.py:420-451 -> int64 indices (B, T)."""
    if mask3 is None:
        mask3 = torch.zeros((x.shape[0], 1, x.shape[1]), dtype=torch.bool)
    z = preencoder_encode_latent(P, x, mask3)
    return fsq_forward(z, levels)[1].long()


def preencoder_decode(P: Params, idx: torch.Tensor, mask3: Optional[torch.Tensor], levels: List[int],
                      return_hidden=False):
    """preencoder.py:453-504 -> x_post [, last_hid]."""
    if mask3 is None:
        mask3 = torch.zeros((idx.shape[0], 1, idx.shape[1]), dtype=torch.bool)
    codes = fsq_indices_to_codes(idx, levels)
    out = preencoder_decode_codes(P, codes, mask3, return_hidden=return_hidden)
    return (out[1], out[2]) if return_hidden else out[1]


# --------------------------------------------------------------------------- discriminators (rows D1, D2)
def _se_masked(P, prefix, x, padded_mask):
    """discriminators.py:40-67."""
    B, C = x.shape[:2]
    valid = ~padded_mask
    denom = valid.sum(dim=(2, 3)).clamp(min=1)
    squeeze = (x * valid).view(B, C, -1).sum(dim=2) / denom
    e = torch.sigmoid(F.linear(F.relu(F.linear(squeeze, P[prefix + "fc1.weight"], P[prefix + "fc1.bias"])),
                               P[prefix + "fc2.weight"], P[prefix + "fc2.bias"]))
    return x * e.view(B, C, 1, 1)


# parity tests: a list here collects, per discriminator layer run with adopted ``gates``, how many adopted branches differ from the
# sign of this implementation's own pre-activation and how large those pre-activations are (see patch_discriminator)
GATE_STATS = None


def patch_discriminator(P: Params, prefix: str, x: torch.Tensor, lengths: torch.Tensor, strides, training: bool,
                        return_features=False, gates=None):
    """discriminators.py:196-242.  x (B, T, F) -> logits (B,1,H,W), patch_mask (True = VALID)[, features].

    ``strides``: per-conv (sh, sw); paddings are (k-1)//2 per dim (:136,159).
    ``gates`` (parity tests only): per layer a bool tensor (B, C, H, W) saying which LeakyReLU branch each element takes
    (True: identity, False: slope) instead of the sign of the pre-activation - the branches another f32 implementation took,
    so that elements within rounding of zero do not decide the comparison.
    """
    B, T, Fq = x.shape
    padded = sequence_mask(T, lengths)[:, None, None, :].expand(-1, 1, Fq, -1)       # :184-193
    out = x.transpose(1, 2).unsqueeze(1)
    n = len(strides)
    feats = []
    for i in range(n):
        if i == n - 1:
            out = _se_masked(P, prefix + "se_block.", out, padded)                    # :216-217
        w = sn_weight(P, prefix + f"convs.{i}.", training)
        kh, kw = w.shape[-2:]
        sh, sw = strides[i]
        inp = out
        out = F.conv2d(out, w, P[prefix + f"convs.{i}.bias"], stride=(sh, sw), padding=((kh - 1) // 2, (kw - 1) // 2))
        pre = out
        out = F.leaky_relu(out, 0.2) if gates is None else torch.where(gates[i], out, 0.2 * out)
        if sh > 1 or sw > 1:
            padded = F.max_pool2d(padded.float(), kernel_size=(sh, sw), stride=(sh, sw), ceil_mode=True).bool()
        if gates is not None and GATE_STATS is not None:
            # how far the adopted branches are from this implementation's own (sign of its own pre-activation), on valid frames:
            # a shared branch may differ only where the pre-activation is within rounding of zero, and only on a handful of elements
            with torch.no_grad():
                valid = ~padded.expand_as(pre)
                differ = ((pre > 0) != gates[i]) & valid & (pre != 0)
                rms = float(pre[valid].double().pow(2).mean().sqrt()) if bool(valid.any()) else 0.0
                # the scale rounding errors of this layer live on: the largest sum of |x| |w| (+ |b|) of an output element (a layer
                # whose terms cancel - the first one, on nearly constant mel frames - has pre-activations far smaller than that)
                terms = F.conv2d(inp.detach().abs(), w.detach().abs(), P[prefix + f"convs.{i}.bias"].detach().abs(), stride=(sh, sw),
                                 padding=((kh - 1) // 2, (kw - 1) // 2))
                GATE_STATS.append({"prefix": prefix, "layer": i, "valid": int(valid.sum()), "differ": int(differ.sum()), "rms": rms,
                                   "terms": float(terms.max()),
                                   "max_abs_pre_where_differ": float(pre[differ].abs().max()) if bool(differ.any()) else 0.0})
        out = out.masked_fill(padded, 0.0)
        if return_features and 2 <= i <= n - 2:                                       # :104-108
            feats.append((out, padded))
    if return_features:
        return out, ~padded, feats
    return out, ~padded


def patch_strides(stride, n_convs):
    """discriminators.py:110-118 plus the final conv's (1,1) (:165)."""
    if isinstance(stride, int):
        ls = [(1, stride)] * n_convs
    elif isinstance(stride, tuple) and len(stride) == 2:
        ls = [stride] * n_convs
    else:
        ls = [tuple(s) for s in stride]
    return ls[:-1] + [(1, 1)]


def multibin_strides(n_convs, n_no_strides):
    """discriminators.py:269-276 plus the final conv's (1,1)."""
    return [(1, 1) if i < n_no_strides else (1, 2) for i in range(n_convs - 1)] + [(1, 1)]


def multibin_discriminator(P: Params, prefix: str, x, lengths, n_bins: int, n_no_strides: int, training: bool,
                           return_features=False, gates=None):
    """discriminators.py:292-311.  ``gates``: as patch_discriminator, every tensor stacked over the bands (n_bins * B, ...)."""
    n_convs = 0
    while f"{prefix}discriminators.0.convs.{n_convs}.bias" in P:
        n_convs += 1
    strides = multibin_strides(n_convs, n_no_strides)
    outs, masks, feats = [], [], []
    for j, sub in enumerate(torch.split(x, x.shape[-1] // n_bins, dim=-1)):
        B = x.shape[0]
        gj = None if gates is None else [g[j * B:(j + 1) * B] for g in gates]
        r = patch_discriminator(P, f"{prefix}discriminators.{j}.", sub, lengths, strides, training, return_features, gj)
        outs.append(r[0])
        masks.append(r[1])
        if return_features:
            feats.append(r[2])
    return (outs, masks, feats) if return_features else (outs, masks)


# --------------------------------------------------------------------------- losses (rows L1-L3)
class LSGANState:
    """EMA state of losses.py:16-19 (one instance is shared by all nine
    discriminators, train.py:216)."""

    def __init__(self, decay=0.99):
        self.decay = decay
        self.ema_real = torch.tensor(0.0)
        self.ema_fake = torch.tensor(0.0)
        self.initialized = False


def _masked_mse(pred, target_value, mask):
    """losses.py:21-35."""
    err = (pred - target_value) ** 2
    if mask is None:
        return err.mean()
    m = mask.float()
    valid = m.sum()
    if valid.item() > 0:
        return (err * m).sum() / valid
    return torch.tensor(0.0)


def lsgan_d_loss(st: LSGANState, real, fake, real_mask, fake_mask, real_label=1.0, fake_label=0.0):
    """losses.py:84-108 with use_lecam=True: 0.5*(mse_real + mse_fake) + lecam."""
    loss = 0.5 * (_masked_mse(real, real_label, real_mask) + _masked_mse(fake, fake_label, fake_mask))
    mr, mf = real_mask.float(), fake_mask.float()
    real_mean = ((real * mr).sum() / mr.sum().clamp(min=1)).detach()               # :37-58
    fake_mean = ((fake * mf).sum() / mf.sum().clamp(min=1)).detach()
    if not st.initialized:
        st.ema_real, st.ema_fake, st.initialized = real_mean.clone(), fake_mean.clone(), True
    else:
        st.ema_real = st.ema_real * st.decay + (1 - st.decay) * real_mean
        st.ema_fake = st.ema_fake * st.decay + (1 - st.decay) * fake_mean
    term_r = (((real - st.ema_fake).clamp(min=0) * mr) ** 2).sum() / mr.sum().clamp(min=1)   # :60-82
    term_f = (((st.ema_real - fake).clamp(min=0) * mf) ** 2).sum() / mf.sum().clamp(min=1)
    return loss + term_r + term_f


def lsgan_g_loss(fake, mask, real_label=1.0):
    """losses.py:110-116."""
    return _masked_mse(fake, real_label, mask)


def masked_mel_loss(x, y, lengths, group_size=1, loss_type="mse", eps=1e-6):
    """losses.py:146-182."""
    B, T, C = x.shape
    G = C // group_size
    mask = sequence_mask(T, lengths).unsqueeze(2).expand(B, T, C).reshape(B, T, G, group_size)
    diff = x.view(B, T, G, group_size) - y.view(B, T, G, group_size)
    per = torch.sqrt(diff ** 2 + eps ** 2) if loss_type == "charbonnier" else diff ** 2
    per = per.masked_fill(mask, 0.0)
    group_sum = per.sum(dim=[0, 1, 3])
    count = (~mask).float().sum(dim=[0, 1, 3])
    return (group_sum / (count + 1e-12)).mean()


def masked_mae(pred, target, mask, eps=1e-8):
    """train.py:38-45."""
    if mask.dim() == 2:
        mask = mask.unsqueeze(1)
    mask = mask.expand_as(pred)
    diff = (pred - target).abs().masked_fill(mask, 0.0)
    return diff.sum() / ((~mask).sum() + eps)


# --------------------------------------------------------------------------- helpers for tests
def params_from_npz(npz, prefix: str, requires_grad=True) -> Params:
    """Pull ``prefix``-ed arrays out of a fixture into a parameter dict.  Float
    tensors become autograd leaves; buffers named weight_u / weight_v do not."""
    P = {}
    names = {k[len(prefix):] for k in npz.files if k.startswith(prefix)}
    for name in sorted(names):
        t = torch.from_numpy(npz[prefix + name].copy())
        if requires_grad and t.is_floating_point() and not is_sn_buffer(name, names):
            t.requires_grad_(True)
        P[name] = t
    return P


def is_sn_buffer(name: str, names) -> bool:
    """spectral-norm u / v buffers (siblings of ``weight_orig``); legacy weight-norm
    also has a ``weight_v`` but that one is a parameter."""
    if name.endswith("weight_u") or name.endswith("weight_v"):
        return (name.rsplit("weight_", 1)[0] + "weight_orig") in names
    return False
