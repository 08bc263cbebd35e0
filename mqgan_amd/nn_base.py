"""Parameter containers with the reference's state_dict key names, producing the effective
weights the implicit-GEMM kernels read.

Three flavours appear in the reference checkpoint layout:
  * new-API weight norm  -> ``<name>.bias``, ``<name>.parametrizations.weight.original0`` (g),
                            ``...original1`` (v)            (preencoder.py:53,248; attentions.py:500-501)
  * legacy weight norm   -> ``<name>.bias``, ``<name>.weight_g``, ``<name>.weight_v``   (attentions.py:469)
  * legacy spectral norm -> ``<name>.bias``, ``<name>.weight_orig`` + buffers ``weight_u``, ``weight_v``
                                                                                (discriminators.py:139,162)
  * plain                -> ``<name>.weight``, ``<name>.bias``
Weights keep the reference's (K, C, *kernel) shapes; the kernels' KRSC / CRSK images are produced on
the fly by mq_weight_prep / mq_spectral_prep (fused with the normalisation).
"""
import math
from typing import Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops


def _fan_in(shape) -> int:
    n = 1
    for s in shape[1:]:
        n *= s
    return n


def _default_init(shape):
    """PyTorch's default Conv/Linear init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
    bound = 1.0 / math.sqrt(_fan_in(shape))
    w = torch.empty(shape).uniform_(-bound, bound)
    b = torch.empty(shape[0]).uniform_(-bound, bound)
    return w, b


class _WNOriginals(nn.Module):
    def __init__(self, g, v):
        super().__init__()
        self.original0 = nn.Parameter(g)
        self.original1 = nn.Parameter(v)


class _Parametrizations(nn.Module):
    def __init__(self, g, v):
        super().__init__()
        self.weight = _WNOriginals(g, v)


class share_effective_weights:
    """``with share_effective_weights(): D(real); D(fake)`` - inside the block every layer's effective weights are
    computed once and reused by later forward passes, as long as the pass does not change them (i.e. no spectral-norm
    power iteration: discriminators in eval mode, which is their state from the first generator step of an epoch
    on, train.py:415-416).  The weight gradient then flows through the normalisation once, with the output
    gradients of all passes already summed - the same derivative the reference accumulates pass by pass."""
    active = None

    def __enter__(self):
        self.prev = share_effective_weights.active
        share_effective_weights.active = {}
        return self

    def __exit__(self, *exc):
        share_effective_weights.active = self.prev
        return False


class ConvParams(nn.Module):
    """Weights of one conv / linear layer.  ``flavour`` in {"plain", "wn", "wn_legacy", "sn"}."""

    def __init__(self, cout: int, cin: int, ksize: Sequence[int], flavour: str = "plain", bias: bool = True,
                 init: str = "default"):
        super().__init__()
        shape = (cout, cin) + tuple(ksize)
        self.flavour = flavour
        self.cout, self.cin, self.ksize = cout, cin, tuple(ksize)
        w, b = _default_init(shape)
        if init == "normal02":                      # discriminators.py:176-181
            w = torch.randn(shape) * 0.02
            b = torch.zeros(cout)
        if flavour == "sn":
            # state_dict order of the reference: bias, weight_orig, then buffers weight_u, weight_v
            self.bias = nn.Parameter(b) if bias else None
            self.weight_orig = nn.Parameter(w)
            u = torch.nn.functional.normalize(torch.randn(cout), dim=0, eps=1e-12)
            v = torch.nn.functional.normalize(torch.randn(_fan_in(shape)), dim=0, eps=1e-12)
            self.register_buffer("weight_u", u)
            self.register_buffer("weight_v", v)
        elif flavour == "wn":
            self.bias = nn.Parameter(b) if bias else None
            g = w.flatten(1).norm(dim=1).view(cout, *([1] * (len(shape) - 1)))
            self.parametrizations = _Parametrizations(g, w)
        elif flavour == "wn_legacy":
            self.bias = nn.Parameter(b) if bias else None
            self.weight_g = nn.Parameter(w.flatten(1).norm(dim=1).view(cout, *([1] * (len(shape) - 1))))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)
            self.bias = nn.Parameter(b) if bias else None

    # effective weights in the kernels' layouts -------------------------------------------------
    def freeze_(self, frozen: bool = True):
        """Inference: compute the effective weights once and reuse them (weights no longer change).
        Never use while training - the optimiser updates parameters in place behind autograd's back."""
        self._frozen = frozen
        self._cache = {}

    def effective(self, need_crsk: bool = True, power_iteration: bool = False):
        if getattr(self, "_frozen", False) and not torch.is_grad_enabled():
            key = (need_crsk,)
            if key not in self._cache:
                self._cache[key] = self._effective(need_crsk, False)
            return self._cache[key]
        shared = share_effective_weights.active
        if shared is not None and not power_iteration:
            key = (id(self), need_crsk)
            if key not in shared:
                hit = shared.get((id(self), True)) if not need_crsk else None     # a CRSK-bearing entry serves both
                shared[key] = hit if hit is not None else self._effective(need_crsk, False)
            return shared[key]
        return self._effective(need_crsk, power_iteration)

    def _effective(self, need_crsk: bool = True, power_iteration: bool = False):
        if self.flavour == "plain":
            return ops.weight_prep(self.weight, None, need_crsk)
        if self.flavour == "wn":
            p = self.parametrizations.weight
            return ops.weight_prep(p.original1, p.original0, need_crsk)
        if self.flavour == "wn_legacy":
            return ops.weight_prep(self.weight_v, self.weight_g, need_crsk)
        return ops.spectral_prep(self.weight_orig, self.weight_u, self.weight_v, power_iteration, need_crsk)

    def fold_weight_norm_(self, both: bool = False):
        """strip_weight_norm (preencoder.py:507-514): the LEGACY flavour's weight_g/weight_v -> weight.  ``both=True`` also
        folds the ``parametrizations`` flavour, which the reference leaves in place (the exported wire format
        carries plain effective weights only)."""
        if self.flavour == "wn_legacy":
            v, g = self.weight_v, self.weight_g
        elif self.flavour == "wn" and both:
            v, g = self.parametrizations.weight.original1, self.parametrizations.weight.original0
        else:
            return
        with torch.no_grad():
            w = v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))
        if self.flavour == "wn_legacy":
            del self.weight_g, self.weight_v
        else:
            del self.parametrizations
        bias = self.bias
        del self.bias
        self.weight = nn.Parameter(w)
        self.bias = bias
        self.flavour = "plain"


def linear(x: torch.Tensor, p: ConvParams, need_dx: bool = True) -> torch.Tensor:
    """nn.Linear on the last dim of a channels-last tensor, as a 1x1 implicit GEMM."""
    cin = x.shape[-1]
    rows = x.numel() // cin
    w_krsc, w_crsk = p.effective(need_crsk=need_dx)
    d = ops.conv_desc(1, 1, rows, cin, p.cout, 1, 1)
    y = ops.conv(x.reshape(1, 1, rows, cin), w_krsc, w_crsk, p.bias, d)
    return y.view(*x.shape[:-1], p.cout)


def conv1d(x: torch.Tensor, p: ConvParams, causal: bool, dilation: int = 1) -> torch.Tensor:
    """x (B, T, Cin) -> (B, T, Cout); "same" padding (odd k) or causal left padding d*(k-1)."""
    B, T, cin = x.shape
    k = p.ksize[0]
    pad = dilation * (k - 1) if causal else dilation * (k // 2)
    w_krsc, w_crsk = p.effective()
    d = ops.conv_desc(B, 1, T, cin, p.cout, 1, k, pad=(0, pad), dil=(1, dilation), out_hw=(1, T))
    return ops.conv(x.view(B, 1, T, cin), w_krsc, w_crsk, p.bias, d).view(B, T, p.cout)


def pad_channels(c: int) -> int:
    """Channel count a tensor of ``c`` channels is STORED with inside the bf16-storage refiner when ops.H2_PAD_CHANNELS is on (it
    is off by default: measured slower, see there): the next multiple of 64 when c is a multiple of 16 above 64 that is not one
    already (hifimusic's 96-channel level -> 128), else c.  The bf16-storage kernels walk the reduction in 64-channel chunks whose
    pixels are whole 128-byte lines; with 96 channels they fall back to 32-channel chunks (half lines, two taps per K-step).  The
    extra channels hold zeros: zero filter rows produce them, APTx(0) = 0 and dropout keep them, zero filter columns ignore them."""
    if ops.ACT_BF16 and ops.H2_PAD_CHANNELS and c > 64 and c % 16 == 0 and c % 64 != 0:
        return (c + 63) // 64 * 64
    return c


def _pad_filters(w_krsc, w_crsk, bias, cin_segments, cout_pad):
    """Zero-padded copies of the effective filters: input channels segment by segment ((real, stored) pairs in tensor order - a
    concatenated input has one pair per part), output channels at the end.  KRSC rows are channel-blocked [C / 16][RS][16] and CRSK
    rows [K / 16][RS][16], so whole 16-channel blocks of zeros go in; autograd slices the padded weight gradient back."""
    K, RS, C = w_krsc.shape
    assert sum(r for r, _ in cin_segments) == C and all(r % 16 == 0 and s % 16 == 0 for r, s in cin_segments if s != r)
    Cp = sum(s for _, s in cin_segments)
    wk = w_krsc.view(K, C // 16, RS, 16) if Cp != C else w_krsc
    if Cp != C:
        parts, pos = [], 0
        for real, stored in cin_segments:
            parts.append(wk[:, pos // 16:(pos + real) // 16])
            if stored > real:
                parts.append(wk.new_zeros(K, (stored - real) // 16, RS, 16))
            pos += real
        wk = torch.cat(parts, 1).reshape(K, RS, Cp)
    if cout_pad > K:
        wk = torch.cat([wk, wk.new_zeros(cout_pad - K, RS, Cp)], 0)
    wc = None
    if w_crsk is not None:
        wc = w_crsk
        if cout_pad > K:
            assert K % 16 == 0 and cout_pad % 16 == 0
            wc = torch.cat([wc.view(C, K // 16, RS, 16), wc.new_zeros(C, (cout_pad - K) // 16, RS, 16)], 1).reshape(C, RS, cout_pad)
        if Cp != C:
            parts, pos = [], 0
            for real, stored in cin_segments:
                parts.append(wc[pos:pos + real])
                if stored > real:
                    parts.append(wc.new_zeros(stored - real, RS, wc.shape[2]))
                pos += real
            wc = torch.cat(parts, 0)
        wc = wc.contiguous()
    b = bias
    if bias is not None and cout_pad > K:
        b = torch.cat([bias, bias.new_zeros(cout_pad - K)])
    return wk.contiguous(), wc, b


def conv2d(x: torch.Tensor, p: ConvParams, stride=(1, 1), pad: Optional[Tuple[int, int]] = None, need_dx: bool = True,
           power_iteration: bool = False, cin_segments=None, cout_pad: Optional[int] = None) -> torch.Tensor:
    """x (N, H, W, Cin) -> (N, P, Q, Cout), padding (k-1)//2 per dim unless given.

    ``cin_segments`` / ``cout_pad`` (pad_channels): x is stored with more channels than the layer has - zeros - and / or the
    output is to be: the filters are zero-padded to match ((real, stored) pairs per part of x, in order)."""
    N, H, W, cin = x.shape
    kh, kw = p.ksize
    if pad is None:
        pad = ((kh - 1) // 2, (kw - 1) // 2)
    w_krsc, w_crsk = p.effective(need_crsk=need_dx, power_iteration=power_iteration)
    bias, cout = p.bias, p.cout
    if cin_segments is not None or (cout_pad is not None and cout_pad != p.cout):
        segs = cin_segments if cin_segments is not None else [(p.cin, p.cin)]
        cout = cout_pad if cout_pad is not None else p.cout
        if any(r != s for r, s in segs) or cout != p.cout:
            w_krsc, w_crsk, bias = _pad_filters(w_krsc, w_crsk, bias, segs, cout)
        assert sum(s for _, s in segs) == cin, (segs, cin)
    d = ops.conv_desc(N, H, W, cin, cout, kh, kw, stride=tuple(stride), pad=pad)
    return ops.conv(x, w_krsc, w_crsk, bias, d)
