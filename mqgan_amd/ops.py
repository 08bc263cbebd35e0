"""Python face of the C ABI: thin wrappers + torch.autograd.Function glue.

PyTorch supplies device memory, the HIP stream and the autograd tape; every
arithmetic step on activations and gradients is a kernel of libmqgan_hip.so.
Layouts are channels-last: (B, T, C) for 1-D maps, (N, H, W, C) for 2-D maps;
frame masks are uint8 (B, T) with 1 = padded.
"""
import ctypes as C
import itertools
import os
from typing import Optional

import torch

from . import _lib
from ._lib import ConvDesc, Cb2dWeights, PointwiseDesc, check, fptr, ptr, stream

F32 = torch.float32


def L():
    lib = _lib.load()
    return lib if PASS_TIMER is None else _PassProxy(lib)


# ----------------------------------------------------------------------------- HBM-bound passes under HIP events (bench.py)
# With a KernelTimer in ``PASS_TIMER`` every launch of the streaming kernels below, and every transform launch inside the
# planned convolution calls (mq_debug_set_pass_hook), is bracketed by two events and recorded under its FAMILY; ``records`` then holds
# (family, 0, start, end) and ``layer_bytes`` the algorithmic activation bytes the reduced-form layers reported.  Off (None) in
# the timed region of the benchmark and everywhere else: L() then returns the library itself.
PASS_TIMER = None
_PASS_FAMILY = {name + suffix: "pointwise_pool_upcat"
                for name in ("mq_pointwise_fwd", "mq_pointwise_bwd", "mq_avgpool_t2_fwd", "mq_avgpool_t2_bwd", "mq_upcat_fwd", "mq_upcat_bwd")
                for suffix in ("", "_h")}
_PASS_FAMILY.update({"mq_fsq_forward": "fsq", "mq_fsq_backward": "fsq"})


class _PassProxy:
    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        fam = _PASS_FAMILY.get(name)
        if fam is None or PASS_TIMER is None:
            return f

        def timed(*a):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            rc = f(*a)
            e.record()
            PASS_TIMER.records.append((fam, 0.0, s, e))
            PASS_TIMER.details.append(name)
            return rc
        return timed


_PASS_STATE = {"start": None}


def _pass_hook(phase, nbytes, family):
    t = PASS_TIMER
    if t is None:
        return
    if phase == 2:
        fam = family.decode() if family else "?"
        t.layer_bytes[fam] = t.layer_bytes.get(fam, 0.0) + nbytes
        return
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    if phase == 0:
        _PASS_STATE["start"] = ev
    else:
        t.records.append((family.decode() if family else "?", 0.0, _PASS_STATE["start"], ev))
        t.details.append(None)


_PASS_HOOK = _lib.PassHook(_pass_hook)             # kept alive for the lifetime of the module


class pass_timing:
    """``with ops.pass_timing() as t: step()`` - HIP events around every HBM-bound pass of the block; ``t.summary()`` after a
    synchronize: family -> launches, ms; ``t.layer_bytes``: form name -> algorithmic activation bytes of the reduced-form layers."""

    def __enter__(self):
        global PASS_TIMER
        t = KernelTimer()
        t.layer_bytes = {}
        _lib.load().mq_debug_set_pass_hook(_PASS_HOOK)
        PASS_TIMER = t
        return t

    def __exit__(self, *exc):
        global PASS_TIMER
        PASS_TIMER = None
        _lib.load().mq_debug_set_pass_hook(_lib.PassHook(0))
        return False


# MQGAN_POISON=1 (debugging): every buffer the kernels are handed "uninitialised" is filled with NaN first, so that an element
# some kernel reads before any kernel wrote it shows up in the results instead of depending on what the allocator had there
_POISON = os.environ.get("MQGAN_POISON") == "1"


def _empty(shape, like, dtype=F32):
    t = torch.empty(shape, dtype=dtype, device=like.device)
    if _POISON and t.is_floating_point():
        t.fill_(float("nan"))
    return t


def _ws_bytes(nbytes, like):
    t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)
    if _POISON:
        t.fill_(0xFF)            # four 0xFF bytes are a NaN
    return t


# ----------------------------------------------------------------------------- dropout seeds
_seed_counter = itertools.count(1)
_base_seed = 0x5DEECE66D


def manual_seed(seed: int, rank: int = 0):
    """Seed the counter-based dropout RNG (per-rank offset for data parallel runs)."""
    global _seed_counter, _base_seed
    _base_seed = (int(seed) * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    _seed_counter = itertools.count(1)


def next_seed() -> int:
    return (_base_seed + next(_seed_counter) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------- convolution
PRECISIONS = {"f32": 0, "bf16": 1, "f32x3": 2}
_conv_precision = 0


def set_conv_precision(name: str) -> str:
    """Matrix engine of every convolution / linear layer created from now on: "f32" (exact products, the parity
    path) or "bf16" (operands rounded to bf16 inside the kernel, f32 accumulate and f32 tensors - the counterpart
    of the reference's autocast region, train.py:330-333).  Returns the previous setting."""
    global _conv_precision
    prev = [k for k, v in PRECISIONS.items() if v == _conv_precision][0]
    _conv_precision = PRECISIONS[name]
    return prev


class conv_precision:
    """``with ops.conv_precision("bf16"): ...`` - scoped set_conv_precision (descriptors capture the setting when they
    are built, so the backward pass of a layer runs on the engine its forward pass used)."""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        self.prev = set_conv_precision(self.name)
        return self

    def __exit__(self, *exc):
        set_conv_precision(self.prev)
        return False


# bf16 STORAGE of activations (training.precision: bf16): inside the refiner and the discriminators every tensor between
# two convolutions is a bf16 tensor - what the reference's autocast region stores (train.py:220, 523) - and the layers
# whose shapes allow it run on the LDS-DMA kernels of csrc/igemm_h2.hip.  Off by default (f32 tensors everywhere).
ACT_BF16 = False
# inside the bf16-storage refiner, tensors whose channel count is a multiple of 16 above 64 but not of 64 CAN be stored zero-padded
# to the next multiple of 64 (nn_base.pad_channels; MQGAN_H2_PAD=1).  Off by default - a recorded negative result: hifimusic's
# 96-channel level as 128 stored channels moves its layers from the 32-channel-chunk kernels (0.43-0.64 PF/s) to the shared-row
# kernels, but at two 64-column tiles those reach only 0.82 PF/s on a third more multiplications: config 3 189.6-189.8 ms/step
# against 187.3-188.4 without (alternating runs in one call, profiles/r5_c3_pad_ab.txt)
H2_PAD_CHANNELS = os.environ.get("MQGAN_H2_PAD", "0") == "1"
# what the caller asked for (Trainer: training.precision == "bf16"); the modules that support bf16 storage (UNetRefiner,
# the discriminators) switch ACT_BF16 on for their own body and hand f32 tensors back at their boundary
ACT_BF16_REQUESTED = False


class request_act_storage:
    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global ACT_BF16_REQUESTED
        self.prev, ACT_BF16_REQUESTED = ACT_BF16_REQUESTED, self.on
        return self

    def __exit__(self, *exc):
        global ACT_BF16_REQUESTED
        ACT_BF16_REQUESTED = self.prev
        return False


class act_storage:
    """``with ops.act_storage(True): ...`` - scoped switch of ACT_BF16."""

    def __init__(self, on: bool):
        self.on = bool(on)

    def __enter__(self):
        global ACT_BF16
        self.prev, ACT_BF16 = ACT_BF16, self.on
        return self

    def __exit__(self, *exc):
        global ACT_BF16
        ACT_BF16 = self.prev
        return False


def conv_desc(N, H, W, Cin, K, R, S, stride=(1, 1), pad=(0, 0), dil=(1, 1), groups=1, out_hw=None,
              precision=None) -> ConvDesc:
    if out_hw is None:
        P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1
        Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
    else:
        P, Q = out_hw
    return ConvDesc(N=N, H=H, W=W, C=Cin, K=K, R=R, S=S, P=P, Q=Q, stride_h=stride[0], stride_w=stride[1], pad_h=pad[0],
                    pad_w=pad[1], dil_h=dil[0], dil_w=dil[1], groups=groups,
                    precision=_conv_precision if precision is None else PRECISIONS.get(precision, precision))


def _desc_key(d: ConvDesc):
    return tuple(getattr(d, f[0]) for f in ConvDesc._fields_)


def _desc_from_key(k) -> ConvDesc:
    return ConvDesc(*k)


class KernelTimer:
    """HIP-event timing of individual kernel launches on the current stream (bench.py's roofline leg).
    Enabled by assigning an instance to ``ops.TIMER``; adds two event records per timed launch."""

    def __init__(self):
        self.records = []          # (tag, algorithmic flops, start event, end event)
        self.details = []

    def run(self, tag, flops, fn, detail=None):
        """``tag`` None: the name of the kernel instantiation the C dispatch launched inside ``fn`` (mq_last_kernel: the
        library says what it ran, nothing here mirrors its choice)."""
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        if tag is None:
            tag = L().mq_last_kernel().decode()
        self.records.append((tag, flops, s, e))
        self.details.append(detail)

    def summary(self):
        """tag -> dict(launches, flops, ms).  Call after torch.cuda.synchronize()."""
        out = {}
        for tag, flops, s, e in self.records:
            d = out.setdefault(tag, {"launches": 0, "flops": 0.0, "ms": 0.0})
            d["launches"] += 1
            d["flops"] += flops
            d["ms"] += s.elapsed_time(e)
        return out


TIMER: Optional[KernelTimer] = None


def _timed(tag, flops, fn, detail=None):
    if TIMER is None:
        fn()
    else:
        TIMER.run(tag, flops, fn, detail)


def _conv_flops(d: ConvDesc):
    return 2.0 * d.N * d.P * d.Q * d.K * d.C * d.R * d.S


def colsum(x2d: torch.Tensor) -> torch.Tensor:
    rows, cols = x2d.shape
    out = _empty((cols,), x2d)
    ws = _empty((1024 * cols,), x2d)
    check(L().mq_colsum(fptr(x2d), rows, cols, fptr(out), fptr(ws), stream()), "mq_colsum")
    return out


BF16 = torch.bfloat16


def cast_bf16(t: torch.Tensor) -> torch.Tensor:
    """bf16 copy (round to nearest even) of an f32 tensor: the operand format of the mq_conv_*_h kernels."""
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=BF16, device=t.device)
    check(L().mq_cast_bf16(fptr(t), ptr(out, BF16), t.numel(), stream()), "mq_cast_bf16")
    return out


def cast_f32(t: torch.Tensor) -> torch.Tensor:
    t = t.contiguous()
    out = torch.empty(t.shape, dtype=F32, device=t.device)
    check(L().mq_cast_f32(ptr(t, BF16), fptr(out), t.numel(), stream()), "mq_cast_f32")
    return out


class CastFn(torch.autograd.Function):
    """dtype change between f32 and bf16 tensors (the gradient is cast back)."""

    @staticmethod
    def forward(ctx, t, to_bf16):
        ctx.to_bf16 = to_bf16
        return cast_bf16(t) if to_bf16 else cast_f32(t)

    @staticmethod
    def backward(ctx, d):
        if d.dtype == (BF16 if ctx.to_bf16 else F32):
            return (cast_f32(d) if ctx.to_bf16 else cast_bf16(d)), None
        return d.contiguous(), None


def to_bf16(t):
    return t if t.dtype == BF16 else CastFn.apply(t, True)


def to_f32(t):
    return t if t.dtype == F32 else CastFn.apply(t, False)


def to_storage(t):
    """Activation tensor in the storage dtype of the current mode (bf16 needs a channel count that is a multiple of 8)."""
    if ACT_BF16 and t.shape[-1] % 8 == 0:
        return to_bf16(t)
    return to_f32(t)


# bf16-storage conv kernels (mq_conv_*_h): 10-17 % faster than rounding inside the kernel and bit-identical, but here the
# bf16 copies are made by a separate cast pass per conv, which costs as much as the kernels gain (measured: 143.8 vs
# 143.0 ms/step at C2).  They pay off once the producing kernels emit the copies; until then they are off by default.
BF16_STORAGE = False


def _h_ok(d: ConvDesc, reduce_channels: int) -> bool:
    """bf16-storage kernels: bf16 engine, reduction channel count a multiple of 16, strides <= 2."""
    return (BF16_STORAGE and d.precision == PRECISIONS["bf16"] and reduce_channels % 16 == 0 and d.stride_h <= 2
            and d.stride_w <= 2)


# ----------------------------------------------------------------------------- bf16-storage convolution (H2 kernels)
def h2_supported(d: ConvDesc) -> bool:
    return bool(L().mq_h2_supported(C.byref(d)))


def h2_wgrad_supported(d: ConvDesc) -> bool:
    return bool(L().mq_h2_wgrad_supported(C.byref(d)))


def h2_weight_image(d: ConvDesc, w: torch.Tensor, dgrad: bool) -> torch.Tensor:
    """bf16 weight image of the H2 kernels from the f32 effective weights (KRSC for fprop, CRSK for dgrad)."""
    n = L().mq_h2_weight_elems(C.byref(d), int(dgrad))
    out = torch.empty((n,), dtype=BF16, device=w.device)
    check(L().mq_h2_weight_image(C.byref(d), int(dgrad), fptr(w), ptr(out, BF16), stream()), "mq_h2_weight_image")
    return out


def conv_fprop_h2(d: ConvDesc, x, w_h2, bias=None, mask=None, dmap=(1, 0, 1), act=0, slope=0.2, out_dtype=None):
    out_dtype = BF16 if out_dtype is None else out_dtype
    y = torch.empty((d.N, d.P, d.Q, d.K), dtype=out_dtype, device=x.device)
    _timed(None, _conv_flops(d), lambda: check(
        L().mq_conv_fprop_h2(C.byref(d), ptr(x, BF16), ptr(w_h2, BF16), fptr(bias, True), ptr(mask, torch.uint8, True), dmap[0],
                             dmap[1], dmap[2], int(act), float(slope), int(out_dtype == F32), ptr(y, out_dtype), stream()),
        "mq_conv_fprop_h2"), _desc_key(d))
    return y


def conv_dgrad_h2(d: ConvDesc, dy, wt_h2, out_dtype=None, gate=None):
    """``gate`` = (x bf16, mask, dmap, slope): x came out of a LeakyReLU + masked_fill layer and dx is returned with respect
    to that layer's pre-activation (mq_conv_dgrad_h2_gated)."""
    out_dtype = BF16 if out_dtype is None else out_dtype
    dx = torch.empty((d.N, d.H, d.W, d.C), dtype=out_dtype, device=dy.device)
    if gate is None:
        _timed(None, _conv_flops(d), lambda: check(
            L().mq_conv_dgrad_h2(C.byref(d), ptr(dy, BF16), ptr(wt_h2, BF16), int(out_dtype == F32), ptr(dx, out_dtype), stream()),
            "mq_conv_dgrad_h2"), _desc_key(d))
    else:
        gx, gmask, gmap, gslope = gate
        gmap = gmap if gmask is not None else (1, 0, 1)
        _timed(None, _conv_flops(d), lambda: check(
            L().mq_conv_dgrad_h2_gated(C.byref(d), ptr(dy, BF16), ptr(wt_h2, BF16), ptr(gx, BF16), ptr(gmask, torch.uint8, True),
                                       gmap[0], gmap[1], gmap[2], float(gslope), int(out_dtype == F32), ptr(dx, out_dtype),
                                       stream()), "mq_conv_dgrad_h2_gated"), _desc_key(d))
    return dx


def conv_wgrad_h2(d: ConvDesc, x, dy, w_shape, want_bias: bool):
    dw = torch.empty(w_shape, dtype=F32, device=x.device)
    db = torch.empty((d.groups * d.K,), dtype=F32, device=x.device) if want_bias else None
    nbytes = L().mq_conv_wgrad_h2_workspace(C.byref(d))
    ws = _ws_bytes(nbytes, x)
    _timed(None, _conv_flops(d), lambda: check(
        L().mq_conv_wgrad_h2(C.byref(d), ptr(x, BF16), ptr(dy, BF16), fptr(dw), fptr(db, True), ptr(ws), nbytes, stream()),
        "mq_conv_wgrad_h2"), _desc_key(d))
    return dw, db


class ConvH2Fn(torch.autograd.Function):
    """Convolution on the bf16-storage kernels: x bf16, y bf16 (or f32), effective weights f32 (``w_krsc`` carries the
    gradient).  ``epi`` = None, or (mask, dmap, slope): LeakyReLU + masked_fill fused into the store (the backward pass
    reads LeakyReLU's derivative off the sign of y).  ``in_gate`` / ``pregated``: see conv_leaky."""

    @staticmethod
    def forward(ctx, x, w_krsc, w_crsk, bias, key, epi, out_f32, in_gate=None, pregated=False):
        d = _desc_from_key(key)
        w_h2 = h2_weight_image(d, w_krsc, False)
        if epi is None:
            y = conv_fprop_h2(d, x, w_h2, bias, out_dtype=F32 if out_f32 else BF16)
            mask = None
        else:
            mask, dmap, slope = epi
            y = conv_fprop_h2(d, x, w_h2, bias, mask=mask, dmap=dmap, act=ACT_LEAKY, slope=slope, out_dtype=F32 if out_f32 else BF16)
        ctx.key, ctx.epi_cfg = key, (None if epi is None else (epi[1], epi[2]))
        ctx.w_shape = tuple(w_krsc.shape)
        ctx.has_bias = bias is not None
        ctx.have_crsk = w_crsk is not None
        ctx.pregated = bool(pregated) and epi is not None
        ctx.in_gate_cfg = None if in_gate is None else (in_gate[1], in_gate[2])
        ctx.save_for_backward(x, w_crsk if w_crsk is not None else w_krsc, y if (epi is not None and not ctx.pregated) else None, mask,
                              None if in_gate is None else in_gate[0])
        return y

    @staticmethod
    def backward(ctx, dout):
        x, w_crsk, y, mask, gate_mask = ctx.saved_tensors
        d = _desc_from_key(ctx.key)
        dout = dout.contiguous()
        if ctx.pregated:                # every consumer handed back the gradient with respect to the pre-activation
            dy = dout
        elif ctx.epi_cfg is not None:
            dmap, slope = ctx.epi_cfg
            if slope != 0.2:
                raise _lib.MqganHipError("the pointwise backward kernel implements LeakyReLU(0.2) only")
            dy = pointwise_bwd_raw(dout if dout.dtype == y.dtype else (cast_bf16(dout) if y.dtype == BF16 else cast_f32(dout)), y,
                                   act=ACT_LEAKY, mask=mask, mask_post=mask is not None, dmap=dmap)
        else:
            dy = dout
        if dy.dtype != BF16:
            dy = cast_bf16(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if not ctx.have_crsk:
                raise _lib.MqganHipError("conv backward wrt the input needs the CRSK weights")
            gate = None if ctx.in_gate_cfg is None else (x, gate_mask, ctx.in_gate_cfg[0], ctx.in_gate_cfg[1])
            dx = conv_dgrad_h2(d, dy, h2_weight_image(d, w_crsk, True), out_dtype=x.dtype, gate=gate)
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[1]:
            if h2_wgrad_supported(d):
                dw, db = conv_wgrad_h2(d, x, dy, ctx.w_shape, want_db)
            else:                       # narrow images (Q < 16): the f32 weight-gradient kernel on f32 copies
                _, dw, db = _conv_backward_f32(d, ctx.key, cast_f32(x), None, cast_f32(dy), ctx.w_shape, False, True, want_db)
        elif want_db:
            db = _bias_grad(d, cast_f32(dy))
        return dx, dw, None, db, None, None, None, None, None


class ConvFn(torch.autograd.Function):
    """y = conv(x, w) + bias on the implicit-GEMM kernels.  ``w_krsc`` (groups*K, R, S, C) carries the
    gradient; ``w_crsk`` is the same weights in the dgrad layout (no gradient of its own)."""

    @staticmethod
    def forward(ctx, x, w_krsc, w_crsk, bias, key):
        d = _desc_from_key(key)
        y = _empty((d.N, d.P, d.Q, d.K), x)
        if _h_ok(d, d.C):
            x_h, w_h = cast_bf16(x), cast_bf16(w_krsc)
            _timed(None, _conv_flops(d), lambda: check(
                L().mq_conv_fprop_h(C.byref(d), ptr(x_h, BF16), ptr(w_h, BF16), fptr(bias, True), fptr(y), stream()),
                "mq_conv_fprop_h"), key)
        else:
            _timed(None, _conv_flops(d), lambda: check(
                L().mq_conv_fprop(C.byref(d), fptr(x), fptr(w_krsc), fptr(bias, True), fptr(y), stream()), "mq_conv_fprop"), key)
        ctx.key = key
        ctx.w_shape = tuple(w_krsc.shape)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w_crsk if w_crsk is not None else w_krsc)
        ctx.have_crsk = w_crsk is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_crsk = ctx.saved_tensors
        dx, dw, db = _conv_backward(ctx, x, w_crsk, dy.contiguous())
        return dx, dw, None, db, None


# ----------------------------------------------------------------------------- reduced-multiplication forms behind the C ABI
# WHICH form a layer takes (direct implicit GEMM; Winograd F(mh x mw, 3x3); the same over the polyphase components of a 5x5 /
# stride-2 layer; F(m, 3) along the rows, possibly nested with F(m', S) along the columns; F(m', S) along the columns) and HOW it
# runs (which transforms, which batch-grouped GEMM, which buffers) is decided and executed in csrc/plan.cpp: mq_conv_plan,
# mq_conv_fprop_planned, mq_conv_backward_planned.  This module holds the switches (module attributes, so that tests and A/B
# runs can change them; their defaults come from the MQGAN_* environment) and ONE autograd Function over the planned calls.
WINO_MIN_CHANNELS = int(os.environ.get("MQGAN_WINO_MINC", "128"))      # 3x3 / stride-1 layers; 0 switches the form off (A/B runs)
WINO_TILE = os.environ.get("MQGAN_WINO_TILE", "auto")                  # auto: fewest products per image; "3", "4", "45", "6"
WINO_KEEP_V_MIB = int(os.environ.get("MQGAN_WINO_KEEP_MIB", "6144"))   # larger transformed inputs are recomputed in backward
WINO_S2D_MIN_CHANNELS = int(os.environ.get("MQGAN_WINO_S2D_MINC", "128"))   # 5x5 / stride-2 layers (polyphase)
WINO1D_MIN_CHANNELS = int(os.environ.get("MQGAN_WINO1D_MINC", "128"))  # (3 x k) layers, F(m, 3) along the rows
WINO1D_TILE = int(os.environ.get("MQGAN_WINO1D_TILE", "0"))            # 0: the tile with the fewest products; 2, 4, 5: fixed
WINOW_MIN_CHANNELS = int(os.environ.get("MQGAN_WINOW_MINC", "128"))    # 3 / 5 / 7 stride-1 taps along the columns

FORM_DIRECT, FORM_WINO2D, FORM_S2D, FORM_WINO1D, FORM_WINOW = _lib.FORM_DIRECT, _lib.FORM_WINO2D, _lib.FORM_S2D, _lib.FORM_WINO1D, _lib.FORM_WINOW
_PLANS = {}


def _plan_cfg():
    tile = str(WINO_TILE)
    if tile not in ("auto", "3", "4", "45", "6"):
        raise _lib.MqganHipError(f"MQGAN_WINO_TILE must be auto, 3, 4, 45 or 6, got {WINO_TILE} (F(2x2, 3x3) left the library in ABI 8)")
    return (int(WINO_MIN_CHANNELS), int(WINO_MIN_CHANNELS != 128 or "MQGAN_WINO_MINC" in os.environ), 0 if tile == "auto" else int(tile),
            int(WINO_S2D_MIN_CHANNELS), int(WINO1D_MIN_CHANNELS), int(WINO1D_TILE), int(WINOW_MIN_CHANNELS), int(WINO_KEEP_V_MIB))


def conv_plan(d: ConvDesc, leaky: bool, filters_only: bool = False) -> "_lib.ConvPlan":
    """mq_conv_plan for this descriptor under the module's current switches (cached): .form, .mh, .mw, .nested, .gemm, .flops,
    .keep_bytes, .ws_fprop, .ws_backward.  ``leaky``: the layer is conv -> LeakyReLU -> masked_fill.  ``filters_only``: the copy of
    the plan whose keep buffer holds the transformed CRSK filters alone (keep_uc_bytes; row / column forms with frozen weights)."""
    cfg = _plan_cfg()
    key = (_desc_key(d), bool(leaky), cfg, bool(filters_only))
    p = _PLANS.get(key)
    if p is None:
        c = _lib.PlanCfg(*cfg)
        p = _lib.ConvPlan()
        check(L().mq_conv_plan(C.byref(d), int(bool(leaky)), C.byref(c), C.byref(p)), "mq_conv_plan")
        if filters_only:
            p.keep_filters_only = 1
        _PLANS[key] = p
    return p


def form_name(p) -> str:
    return L().mq_form_name(p.form).decode()


# thin predicates over the plan (tests, tools)
def _wino_ok(d: ConvDesc) -> bool:
    return conv_plan(d, False).form == FORM_WINO2D


def _wino_s2d_ok(d: ConvDesc) -> bool:
    return conv_plan(d, True).form == FORM_S2D


def _wino1d_ok(d: ConvDesc) -> bool:
    return conv_plan(d, True).form == FORM_WINO1D


def _winow_ok(d: ConvDesc) -> bool:
    return conv_plan(d, False).form == FORM_WINOW or conv_plan(d, True).form == FORM_WINOW


# HIP-event timing of the GEMM launches inside the planned calls: the library calls back around each of them (mq_debug_set_launch_hook)
_HOOK_STATE = {"start": None, "detail": None}


def _launch_hook(phase, flops, kernel):
    t = TIMER
    if t is None:
        return
    if phase == 0:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        _HOOK_STATE["start"] = ev
    else:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        t.records.append((kernel.decode() if kernel else "?", flops, _HOOK_STATE["start"], ev))
        t.details.append(_HOOK_STATE["detail"])


_LAUNCH_HOOK = _lib.LaunchHook(_launch_hook)       # kept alive for the lifetime of the module
_hook_installed = False


def _planned_call(key, fn):
    """Run a planned entry point; with a KernelTimer active the library reports every GEMM launch to it."""
    global _hook_installed
    if TIMER is not None and not _hook_installed:
        L().mq_debug_set_launch_hook(_LAUNCH_HOOK)
        _hook_installed = True
    _HOOK_STATE["detail"] = key
    fn()


class ConvPlannedFn(torch.autograd.Function):
    """One layer in the form mq_conv_plan chose for it, forward by mq_conv_fprop_planned and backward by
    mq_conv_backward_planned (csrc/plan.cpp).  ``leaky`` None: y = conv(x, w) + bias.  ``leaky`` = (mask, dmap, slope): y = padded ?
    0 : LeakyReLU(conv + bias) with the discriminators' gate contract - ``in_gate`` = (mask, dmap, slope) of the layer that produced
    x and handed it out pregated (this layer's input gradient then carries that layer's LeakyReLU' and mask); ``pregated``: every
    consumer of y applies LeakyReLU'(y) and the mask itself, so the incoming gradient is already the one with respect to the
    pre-activation.  The transformed input (and the CRSK transformed filters) travel from forward to backward in the plan's keep
    buffer when the weights need a gradient."""

    @staticmethod
    def forward(ctx, x, w_krsc, w_crsk, bias, key, leaky=None, in_gate=None, pregated=False):
        d = _desc_from_key(key)
        p = conv_plan(d, leaky is not None)
        x, w_krsc = x.contiguous(), w_krsc.contiguous()
        mask, dmap, slope = (leaky[0], tuple(leaky[1]), float(leaky[2])) if leaky is not None else (None, (1, 0, 1), 0.0)
        # the keep buffer carries the transformed input (weight gradient) and, for the row / column forms, the CRSK transformed
        # filters (input gradient); the two-dimensional forms transform the rotated filter in backward instead.  With frozen
        # weights (the generator step's passes through the discriminators) only those filters are kept (the plan's filters-only
        # copy): the transformed input - 1.5x the layer's input and more - is not carried from forward to backward for nothing
        need_keep = p.keep_bytes > 0 and ctx.needs_input_grad[1]
        keep = _ws_bytes(p.keep_bytes, x) if need_keep else None
        if not need_keep and ctx.needs_input_grad[0] and p.keep_uc_bytes > 0:
            p = conv_plan(d, leaky is not None, filters_only=True)
            keep = _ws_bytes(p.keep_uc_bytes, x)
        ws = _ws_bytes(p.ws_fprop, x) if p.ws_fprop else None
        y = _empty((d.N, d.P, d.Q, d.K), x)
        _planned_call(key, lambda: check(
            L().mq_conv_fprop_planned(C.byref(p), fptr(x), fptr(w_krsc), fptr(bias, True), 2 if leaky is not None else 0, slope,
                                      ptr(mask, torch.uint8, True), dmap[0], dmap[1], dmap[2], fptr(y), ptr(keep, torch.uint8, True),
                                      ptr(ws, torch.uint8, True), p.ws_fprop, stream()), "mq_conv_fprop_planned"))
        ctx.key, ctx.leaky_cfg = key, (None if leaky is None else (dmap, slope))
        ctx.w_shape = tuple(w_krsc.shape)
        ctx.has_bias = bias is not None
        ctx.pregated = bool(pregated)
        ctx.in_gate = None if in_gate is None else (tuple(in_gate[1]), float(in_gate[2]))
        direct = p.form == FORM_DIRECT
        ctx.plan = p       # backward runs the plan forward ran with, whatever the module's switches say by then
        # what backward reads: x for the direct weight gradient, as the gate tensor, or to transform again without a keep buffer
        need_x = in_gate is not None or (ctx.needs_input_grad[1] and (direct or keep is None or p.keep_filters_only))
        ctx.save_for_backward(x if need_x else None, keep, w_krsc, w_crsk if direct else None,
                              y if (leaky is not None and not pregated) else None, mask, None if in_gate is None else in_gate[0])
        return y

    @staticmethod
    def backward(ctx, dout):
        x, keep, w_krsc, w_crsk, y, mask, mask_in = ctx.saved_tensors
        d = _desc_from_key(ctx.key)
        p = ctx.plan
        dy = dout.contiguous()
        if ctx.leaky_cfg is not None and not ctx.pregated:
            dmap, slope = ctx.leaky_cfg
            if slope != 0.2:
                raise _lib.MqganHipError("the pointwise backward kernel implements LeakyReLU(0.2) only")
            Cc = y.shape[-1]
            pd = PointwiseDesc(rows=y.numel() // Cc, C=Cc, act=ACT_LEAKY, mask_pre=0, mask_post=int(mask is not None),
                               d1=dmap[0], d2=dmap[1], d3=dmap[2], p_drop=0.0, seed=0)
            dh = torch.empty_like(y)
            check(L().mq_pointwise_bwd(C.byref(pd), fptr(dy), fptr(y), None, ptr(mask, torch.uint8, True), None,
                                       None, fptr(dh), None, None, None, None, stream()), "mq_pointwise_bwd")
            dy = dh
        want_dx, want_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        want_db = ctx.has_bias and ctx.needs_input_grad[3]
        if want_dx and p.form == FORM_DIRECT and w_crsk is None:
            raise _lib.MqganHipError("conv backward wrt the input needs the CRSK weights")
        if want_dw and keep is None and x is None:
            raise _lib.MqganHipError("ConvPlannedFn: weight gradient asked for a layer whose weights did not require one")
        dx = _empty((d.N, d.H, d.W, d.C), dy) if want_dx else None
        dw = _empty(ctx.w_shape, dy) if want_dw else None
        db = _empty((d.groups * d.K,), dy) if want_db else None
        gmask, gdmap, gslope = (mask_in, ctx.in_gate[0], ctx.in_gate[1]) if ctx.in_gate is not None else (None, (1, 0, 1), 1.0)
        ws = _ws_bytes(p.ws_backward, dy) if p.ws_backward else None
        _planned_call(ctx.key, lambda: check(
            L().mq_conv_backward_planned(C.byref(p), fptr(x, True), ptr(keep, torch.uint8, True), fptr(w_krsc), fptr(w_crsk, True), fptr(dy),
                                         fptr(x, True) if ctx.in_gate is not None else None, gslope, ptr(gmask, torch.uint8, True),
                                         gdmap[0], gdmap[1], gdmap[2], fptr(dx, True), fptr(dw, True), fptr(db, True),
                                         ptr(ws, torch.uint8, True), p.ws_backward, stream()), "mq_conv_backward_planned"))
        return dx, dw, None, db, None, None, None, None


def _bias_grad(d, dy):
    rows_per_group = (d.N // d.groups) * d.P * d.Q
    if d.groups == 1:
        return colsum(dy.view(-1, d.K))
    return torch.cat([colsum(dy.view(d.groups, rows_per_group, d.K)[g]) for g in range(d.groups)])


def _conv_backward(ctx, x, w_crsk, dy):
    """dgrad / wgrad (+ fused bias gradient) of a conv whose forward stored key, w_shape, has_bias, have_crsk on ctx;
    inputs 0, 1, 3 of the Function are x, w_krsc, bias."""
    if ctx.needs_input_grad[0] and not ctx.have_crsk:
        raise _lib.MqganHipError("conv backward wrt the input needs the CRSK weights")
    d = _desc_from_key(ctx.key)
    return _conv_backward_f32(d, ctx.key, x, w_crsk, dy, ctx.w_shape, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                              ctx.has_bias and ctx.needs_input_grad[3], in_gate=getattr(ctx, "in_gate", None))


def _conv_backward_f32(d, key, x, w_crsk, dy, w_shape, want_dx, want_dw, want_db, in_gate=None):
    """``in_gate`` = (mask, dmap, slope): x came out of a LeakyReLU + masked_fill layer that expects its gradient with respect
    to the pre-activation - fused into the dgrad store where the engine has the epilogue, one pointwise pass otherwise."""
    class _C:       # the body below was written against an autograd ctx
        pass
    ctx = _C()
    ctx.key, ctx.w_shape = key, w_shape
    ctx.needs_input_grad = (want_dx, want_dw, False, want_db)
    ctx.has_bias = want_db
    dx = dw = db = None
    if ctx.needs_input_grad[0]:
        dx = torch.empty_like(x)
        if _h_ok(d, d.K):
            dy_h, wc_h = cast_bf16(dy), cast_bf16(w_crsk)
            _timed(None, _conv_flops(d), lambda: check(
                L().mq_conv_dgrad_h(C.byref(d), ptr(dy_h, BF16), ptr(wc_h, BF16), fptr(dx), stream()), "mq_conv_dgrad_h"),
                ctx.key)
            if in_gate is not None:
                dx = pointwise_bwd_raw(dx, x, act=ACT_LEAKY, mask=in_gate[0], mask_post=in_gate[0] is not None, dmap=in_gate[1])
        elif in_gate is not None:
            gmask, gmap, gslope = in_gate
            gmap = gmap if gmask is not None else (1, 0, 1)
            _timed(None, _conv_flops(d), lambda: check(
                L().mq_conv_dgrad_gated(C.byref(d), fptr(dy), fptr(w_crsk), fptr(x), ptr(gmask, torch.uint8, True), gmap[0], gmap[1],
                                        gmap[2], float(gslope), fptr(dx), stream()), "mq_conv_dgrad_gated"), ctx.key)
        else:
            _timed(None, _conv_flops(d), lambda: check(
                L().mq_conv_dgrad(C.byref(d), fptr(dy), fptr(w_crsk), fptr(dx), stream()), "mq_conv_dgrad"), ctx.key)
    want_db = ctx.has_bias and ctx.needs_input_grad[3]
    if ctx.needs_input_grad[1]:
        dw = _empty(ctx.w_shape, x)      # (groups*K, R*S, C) == KRSC
        if want_db:                      # bias gradient rides on the dy tiles of the weight-gradient GEMM
            db = _empty((d.groups * d.K,), x)
            nbytes = L().mq_conv_wgrad_bias_workspace(C.byref(d))
            ws = _ws_bytes(nbytes, x)
            _timed(None, _conv_flops(d), lambda: check(
                L().mq_conv_wgrad_bias(C.byref(d), fptr(x), fptr(dy), fptr(dw), fptr(db), ptr(ws), nbytes, stream()),
                "mq_conv_wgrad_bias"), ctx.key)
        else:
            nbytes = L().mq_conv_wgrad_workspace(C.byref(d))
            ws = _ws_bytes(nbytes, x)
            _timed(None, _conv_flops(d), lambda: check(
                L().mq_conv_wgrad(C.byref(d), fptr(x), fptr(dy), fptr(dw), ptr(ws), nbytes, stream()), "mq_conv_wgrad"),
                ctx.key)
    elif want_db:
        db = _bias_grad(d, dy)
    return dx, dw, db


class ConvLeakyFn(torch.autograd.Function):
    """y = padded ? 0 : LeakyReLU(conv(x, w) + bias), activation and mask fused into the conv's store
    (mq_conv_fprop_leaky).  The pre-activation is never written: the backward pass reads LeakyReLU's derivative off
    the sign of y.  ``in_gate`` / ``pregated``: see conv_leaky."""

    @staticmethod
    def forward(ctx, x, w_krsc, w_crsk, bias, mask, key, dmap, slope, in_gate=None, pregated=False):
        d = _desc_from_key(key)
        y = _empty((d.N, d.P, d.Q, d.K), x)
        _timed(None, _conv_flops(d), lambda: check(
            L().mq_conv_fprop_leaky(C.byref(d), fptr(x), fptr(w_krsc), fptr(bias, True), ptr(mask, torch.uint8, True), dmap[0],
                                    dmap[1], dmap[2], float(slope), fptr(y), stream()), "mq_conv_fprop_leaky"), key)
        ctx.key, ctx.dmap, ctx.slope = key, dmap, slope
        ctx.w_shape = tuple(w_krsc.shape)
        ctx.has_bias = bias is not None
        ctx.have_crsk = w_crsk is not None
        ctx.pregated = bool(pregated)
        ctx.in_gate = None if in_gate is None else (in_gate[0], in_gate[1], in_gate[2])
        ctx.save_for_backward(x, w_crsk if w_crsk is not None else w_krsc, None if ctx.pregated else y, mask)
        return y

    @staticmethod
    def backward(ctx, dout):
        x, w_crsk, y, mask = ctx.saved_tensors
        if ctx.slope != 0.2:
            raise _lib.MqganHipError("the pointwise backward kernel implements LeakyReLU(0.2) only")
        if ctx.pregated:                # every consumer handed back the gradient with respect to the pre-activation
            dy = dout.contiguous()
        else:
            Cc = y.shape[-1]
            pd = PointwiseDesc(rows=y.numel() // Cc, C=Cc, act=ACT_LEAKY, mask_pre=0, mask_post=int(mask is not None),
                               d1=ctx.dmap[0], d2=ctx.dmap[1], d3=ctx.dmap[2], p_drop=0.0, seed=0)
            dy = torch.empty_like(y)
            check(L().mq_pointwise_bwd(C.byref(pd), fptr(dout.contiguous()), fptr(y), None, ptr(mask, torch.uint8, True), None,
                                       None, fptr(dy), None, None, None, None, stream()), "mq_pointwise_bwd")
        dx, dw, db = _conv_backward(ctx, x, w_crsk, dy)
        return dx, dw, None, db, None, None, None, None, None, None


class LeakyGateFn(torch.autograd.Function):
    """Identity on a tensor that came out of a ``pregated`` LeakyReLU layer, for a consumer without a gated backward of its
    own: the gradient passing through is multiplied by LeakyReLU'(x) and zeroed on padded frames (one pointwise pass)."""

    @staticmethod
    def forward(ctx, x, mask, dmap):
        ctx.dmap = dmap
        ctx.save_for_backward(x, mask)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dout):
        x, mask = ctx.saved_tensors
        dout = dout.contiguous()
        if dout.dtype != x.dtype:
            dout = cast_bf16(dout) if x.dtype == BF16 else cast_f32(dout)
        return pointwise_bwd_raw(dout, x, act=ACT_LEAKY, mask=mask, mask_post=mask is not None, dmap=ctx.dmap), None, None


class TapsGatherFn(torch.autograd.Function):
    """img (N,H,W) -> cols (N,P,Q,T): the taps of every output pixel of a single-input-channel conv."""

    @staticmethod
    def forward(ctx, img, key, T):
        d = _desc_from_key(key)
        cols = _empty((d.N, d.P, d.Q, T), img)
        check(L().mq_taps_gather(C.byref(d), T, fptr(img), fptr(cols), stream()), "mq_taps_gather")
        ctx.key, ctx.T = key, T
        return cols

    @staticmethod
    def backward(ctx, dcols):
        d = _desc_from_key(ctx.key)
        dimg = _empty((d.N, d.H, d.W), dcols)
        check(L().mq_taps_scatter(C.byref(d), ctx.T, fptr(dcols.contiguous()), fptr(dimg), stream()), "mq_taps_scatter")
        return dimg, None, None


class TapsReduceFn(torch.autograd.Function):
    """u (N,H,W,T) -> y (N,P,Q): folds the per-tap partial products of a single-output-channel conv, adds the bias."""

    @staticmethod
    def forward(ctx, u, bias, key, T):
        d = _desc_from_key(key)
        y = _empty((d.N, d.P, d.Q), u)
        check(L().mq_taps_reduce(C.byref(d), T, fptr(u), fptr(bias, True), fptr(y), stream()), "mq_taps_reduce")
        ctx.key, ctx.T, ctx.has_bias = key, T, bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        d = _desc_from_key(ctx.key)
        dy = dy.contiguous()
        du = _empty((d.N, d.H, d.W, ctx.T), dy)
        check(L().mq_taps_expand(C.byref(d), ctx.T, fptr(dy), fptr(du), stream()), "mq_taps_expand")
        db = None
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = torch.cat([colsum(dy.view(d.groups, -1, 1)[g]) for g in range(d.groups)]) if d.groups > 1 else colsum(dy.view(-1, 1))
        return du, db, None, None


def _thin_conv(x, w_krsc, w_crsk, bias, desc: ConvDesc, leaky=None, out_f32=True, pregate=False, chain=False):
    """C = 1 or K = 1 with more than one tap (csrc/taps.hip): tap gather / fold around a 1x1 GEMM.  With bf16 activation
    storage the tap axis is padded to a multiple of 32 so that the 1x1 GEMM (the only pass over the wide tensor) runs on
    the bf16-storage kernels and the wide tensor is a bf16 tensor from the start."""
    G, K, Cin, RS = desc.groups, desc.K, desc.C, desc.R * desc.S
    storage = ACT_BF16 and desc.precision == PRECISIONS["bf16"]
    T = (RS + 31) // 32 * 32 if storage else (RS + 3) // 4 * 4
    key = _desc_key(desc)
    need_dx = x.requires_grad
    if Cin == 1:
        cols = TapsGatherFn.apply(x.view(desc.N, desc.H, desc.W), key, T)
        wk = torch.nn.functional.pad(w_krsc.view(G * K, RS), (0, T - RS)).contiguous().view(G * K, 1, T)         # KRSC of the 1x1 conv
        wc = wk.view(G, K, T).transpose(1, 2).contiguous().view(G * T, 1, K) if need_dx else None              # its CRSK
        d1 = conv_desc(desc.N, desc.P, desc.Q, T, K, 1, 1, groups=G, precision=desc.precision)
        if _use_h2(d1):
            epi = None if leaky is None else (leaky[0], leaky[1], leaky[2])
            y = ConvH2Fn.apply(to_bf16(cols), wk, wc, bias, _desc_key(d1), epi, False, None, pregate and epi is not None)
            return (y, pregate and epi is not None) if chain else y
        if leaky is not None and not _h_ok(d1, T):          # same output rows (n, p, q): the fused activation applies
            y = ConvLeakyFn.apply(cols, wk, wc, bias, leaky[0], _desc_key(d1), leaky[1], leaky[2], None, pregate)
            return (y, bool(pregate)) if chain else y
        y = ConvFn.apply(cols, wk, wc, bias, _desc_key(d1))
        if leaky is not None:
            y = pointwise(y, act=ACT_LEAKY, mask=leaky[0], mask_post=leaky[0] is not None, dmap=leaky[1])
        return (y, False) if chain else y
    # K = 1: w_krsc is (G, RS, C) in the channel-blocked order [C/CB][RS][CB] -> rows = taps, plain channels
    CB = 16 if Cin % 16 == 0 else Cin
    wt = w_krsc.view(G, Cin // CB, RS, CB).permute(0, 2, 1, 3).contiguous().view(G, RS, Cin)
    wk = torch.nn.functional.pad(wt, (0, 0, 0, T - RS)).contiguous().view(G * T, 1, Cin)                        # KRSC: T "output channels"
    wc = wk.view(G, T, Cin).transpose(1, 2).contiguous().view(G * Cin, 1, T) if need_dx else None
    d1 = conv_desc(desc.N, desc.H, desc.W, Cin, T, 1, 1, groups=G, precision=desc.precision)
    if _use_h2(d1):
        u = ConvH2Fn.apply(to_bf16(x), wk, wc, None, _desc_key(d1), None, True)                                 # u: f32 (N, H, W, T)
    else:
        u = ConvFn.apply(to_f32(x), wk, wc, None, _desc_key(d1))
    return TapsReduceFn.apply(u, bias, key, T).view(desc.N, desc.P, desc.Q, 1)


def _use_h2(desc: ConvDesc) -> bool:
    # (tiny GEMMs - the squeeze-excite linears - stay on the f32 path: a 256-row tile would be mostly padding)
    return (ACT_BF16 and desc.precision == PRECISIONS["bf16"] and desc.N * desc.P * desc.Q >= 2048 and h2_supported(desc))


def _is_thin(desc: ConvDesc) -> bool:
    return desc.R * desc.S > 1 and ((desc.C == 1) != (desc.K == 1)) and (desc.C % 4 == 0 or desc.K % 4 == 0)


def conv_leaky(x, w_krsc, w_crsk, bias, desc: ConvDesc, mask, dmap, slope: float = 0.2, out_f32: bool = False, in_gate=None,
               pregate: bool = False, chain: bool = False):
    """LeakyReLU(conv) with padded frames zeroed - one kernel where the fused epilogue applies, conv + pointwise otherwise.

    Chains of such layers (the discriminators) can drop the stand-alone LeakyReLU backward passes:
    ``pregate=True`` asks for an output whose incoming gradient is taken to be the one with respect to the layer's
    PRE-activation - every consumer of that tensor then has to apply LeakyReLU'(y) and the mask itself.  With ``chain=True``
    the call returns (y, honoured); when honoured, pass ``in_gate=(mask, dmap, slope)`` of this layer to the conv_leaky that
    consumes y (its dgrad kernel applies the gate in its store), give ``leaky_slope`` to masked_mae, or wrap y in LeakyGateFn."""
    if pregate and not chain:
        raise _lib.MqganHipError("conv_leaky(pregate=True) needs chain=True: the caller must know whether the gate was honoured")

    def ret(y, honoured=False):
        return (y, bool(honoured)) if chain else y
    if _use_h2(desc):
        return ret(ConvH2Fn.apply(to_bf16(x), w_krsc, w_crsk, bias, _desc_key(desc), (mask, tuple(dmap), slope), out_f32, in_gate,
                                  pregate), pregate)
    thin = _is_thin(desc)
    if in_gate is not None and (thin or desc.C % 4 != 0 or _h_ok(desc, desc.C)):
        x = LeakyGateFn.apply(x, in_gate[0], tuple(in_gate[1]))         # no gated dgrad on these paths: one pointwise pass
        in_gate = None
    if thin and desc.C == 1:
        return _thin_conv(to_f32(x), w_krsc, w_crsk, bias, desc, leaky=(mask, tuple(dmap), slope), pregate=pregate, chain=chain)
    if thin:
        return ret(pointwise(_thin_conv(x, w_krsc, w_crsk, bias, desc), act=ACT_LEAKY, mask=mask, mask_post=mask is not None, dmap=dmap))
    if x.dtype != F32 and in_gate is not None:
        x = LeakyGateFn.apply(x, in_gate[0], tuple(in_gate[1]))         # the gate tensor of the f32 kernels is the f32 input itself
        in_gate = None
    x = to_f32(x)
    if desc.C % 4 != 0 or _h_ok(desc, desc.C):
        return ret(pointwise(conv(x, w_krsc, w_crsk, bias, desc), act=ACT_LEAKY, mask=mask, mask_post=mask is not None, dmap=dmap))
    if conv_plan(desc, True).form != FORM_DIRECT:          # F(m, 3) along the rows, polyphase F(mh x mw, 3x3), F(m, S) along the columns
        return ret(ConvPlannedFn.apply(x, w_krsc, None, bias, _desc_key(desc), (mask, tuple(dmap), slope), in_gate, pregate), pregate)
    return ret(ConvLeakyFn.apply(x, w_krsc, w_crsk, bias, mask, _desc_key(desc), tuple(dmap), slope, in_gate, pregate), pregate)


def conv(x, w_krsc, w_crsk, bias, desc: ConvDesc, out_f32: bool = False):
    """Convolution on the engine the descriptor names.  With bf16 activation storage on, layers the H2 kernels take read
    and write bf16 tensors (f32 output on request); every other layer reads and writes f32 tensors."""
    if _use_h2(desc):
        return ConvH2Fn.apply(to_bf16(x), w_krsc, w_crsk, bias, _desc_key(desc), None, out_f32)
    if _is_thin(desc):
        return _thin_conv(x if desc.K == 1 else to_f32(x), w_krsc, w_crsk, bias, desc)
    if conv_plan(desc, False).form != FORM_DIRECT:         # F(mh x mw, 3x3) or F(m, S) along the columns
        return ConvPlannedFn.apply(to_f32(x), w_krsc, None, bias, _desc_key(desc))
    return ConvFn.apply(to_f32(x), w_krsc, w_crsk, bias, _desc_key(desc))


# ----------------------------------------------------------------------------- weight preparation
class WeightPrepFn(torch.autograd.Function):
    """(v (K, C, *ks), g (K,...) or None) -> (w_krsc (K, RS, C), w_crsk (C, RS, K)).  mode 1 = weight norm."""

    @staticmethod
    def forward(ctx, v, g, need_crsk):
        K, Cin = v.shape[0], v.shape[1]
        RS = v.numel() // (K * Cin)
        mode = 0 if g is None else 1
        w_krsc = _empty((K, RS, Cin), v)
        w_crsk = _empty((Cin, RS, K), v) if need_crsk else None
        inv = _empty((K,), v) if mode else None
        check(L().mq_weight_prep(mode, fptr(v), fptr(g, True), K, Cin, RS, fptr(w_krsc), fptr(w_crsk, True), fptr(inv, True),
                                 stream()), "mq_weight_prep")
        ctx.mode, ctx.dims = mode, (K, Cin, RS)
        ctx.save_for_backward(v, g, inv)
        ctx.set_materialize_grads(False)       # no zero tensor for the CRSK copy's (absent) gradient
        if w_crsk is not None:
            ctx.mark_non_differentiable(w_crsk)
        return w_krsc, w_crsk

    @staticmethod
    def backward(ctx, dw_krsc, _unused):
        if dw_krsc is None:
            return None, None, None
        v, g, inv = ctx.saved_tensors
        K, Cin, RS = ctx.dims
        dv = torch.empty_like(v)
        dg = torch.empty_like(g) if ctx.mode else None
        check(L().mq_weight_prep_bwd(ctx.mode, fptr(dw_krsc.contiguous()), fptr(v), fptr(g, True), fptr(inv, True), K, Cin, RS,
                                     fptr(dv), fptr(dg, True), stream()), "mq_weight_prep_bwd")
        return dv, dg, None


def weight_prep(v, g=None, need_crsk=True):
    return WeightPrepFn.apply(v, g, need_crsk)


class SpectralPrepFn(torch.autograd.Function):
    """Legacy spectral norm: (weight_orig, u, v) -> (w_krsc, w_crsk); u, v updated in place when
    ``power_iteration``."""

    @staticmethod
    def forward(ctx, w_orig, u, v, power_iteration, need_crsk):
        K, Cin = w_orig.shape[0], w_orig.shape[1]
        RS = w_orig.numel() // (K * Cin)
        w_krsc = _empty((K, RS, Cin), w_orig)
        w_crsk = _empty((Cin, RS, K), w_orig) if need_crsk else None
        sigma = _empty((1,), w_orig)
        scratch = _empty((K + Cin * RS + 8,), w_orig)
        check(L().mq_spectral_prep(fptr(w_orig), fptr(u), fptr(v), K, Cin, RS, int(bool(power_iteration)), 1e-12, fptr(w_krsc),
                                   fptr(w_crsk, True), fptr(sigma), fptr(scratch), stream()), "mq_spectral_prep")
        ctx.dims = (K, Cin, RS)
        # u, v as used for sigma (after the power iteration); clone so later iterations do not disturb backward
        ctx.save_for_backward(w_orig, u.clone() if power_iteration else u, v.clone() if power_iteration else v, sigma)
        ctx.set_materialize_grads(False)       # no zero tensor for the CRSK copy's (absent) gradient
        if w_crsk is not None:
            ctx.mark_non_differentiable(w_crsk)
        return w_krsc, w_crsk

    @staticmethod
    def backward(ctx, dw_krsc, _unused):
        if dw_krsc is None:
            return None, None, None, None, None
        w_orig, u, v, sigma = ctx.saved_tensors
        K, Cin, RS = ctx.dims
        dw = torch.empty_like(w_orig)
        scratch = _empty((264,), w_orig)
        check(L().mq_spectral_prep_bwd(fptr(dw_krsc.contiguous()), fptr(w_orig), fptr(u), fptr(v), fptr(sigma), K, Cin, RS,
                                       fptr(dw), fptr(scratch), stream()), "mq_spectral_prep_bwd")
        return dw, None, None, None, None


def spectral_prep(w_orig, u, v, power_iteration, need_crsk=True):
    return SpectralPrepFn.apply(w_orig, u, v, power_iteration, need_crsk)


SN_MAX_BANDS = 16


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class SpectralPrepBandsFn(torch.autograd.Function):
    """SpectralPrepFn (no power iteration) for the same layer of G band discriminators at once: G weight_orig tensors ->
    stacked (G*K, RS, C) / (G*C, RS, K) filters with two launches instead of 3 G (and no concatenation), backward with two
    instead of 2 G."""

    @staticmethod
    def forward(ctx, need_crsk, us, vs, *w_origs):
        G = len(w_origs)
        w0 = w_origs[0]
        K, Cin = w0.shape[0], w0.shape[1]
        RS = w0.numel() // (K * Cin)
        for t in tuple(w_origs) + tuple(us) + tuple(vs):
            if t.dtype != F32 or not t.is_contiguous() or t.device != w0.device:
                raise _lib.MqganHipError("spectral_prep_bands: f32 contiguous tensors on one device expected")
        if any(tuple(w.shape) != tuple(w0.shape) for w in w_origs):
            raise _lib.MqganHipError("spectral_prep_bands: the bands' weights differ in shape")
        w_krsc = _empty((G * K, RS, Cin), w0)
        w_crsk = _empty((G * Cin, RS, K), w0) if need_crsk else None
        sigma = _empty((G,), w0)
        scratch = _empty((G * max(K, 256),), w0)
        check(L().mq_spectral_prep_bands(_ptr_array(w_origs), _ptr_array(us), _ptr_array(vs), G, K, Cin, RS, fptr(w_krsc),
                                         fptr(w_crsk, True), fptr(sigma), fptr(scratch), stream()), "mq_spectral_prep_bands")
        ctx.dims = (G, K, Cin, RS)
        ctx.uv = (tuple(us), tuple(vs))                 # buffers: untouched without power iteration
        ctx.save_for_backward(sigma, *w_origs)
        ctx.set_materialize_grads(False)
        if w_crsk is not None:
            ctx.mark_non_differentiable(w_crsk)
        return w_krsc, w_crsk

    @staticmethod
    def backward(ctx, dw_krsc, _unused):
        G, K, Cin, RS = ctx.dims
        if dw_krsc is None:
            return (None,) * (3 + G)
        sigma, *w_origs = ctx.saved_tensors
        us, vs = ctx.uv
        dws = [torch.empty_like(w) for w in w_origs]
        scratch = _empty((G * 256,), sigma)
        check(L().mq_spectral_prep_bands_bwd(fptr(dw_krsc.contiguous()), _ptr_array(w_origs), _ptr_array(us), _ptr_array(vs), fptr(sigma),
                                             G, K, Cin, RS, _ptr_array(dws), fptr(scratch), stream()), "mq_spectral_prep_bands_bwd")
        return (None, None, None) + tuple(dws)


def spectral_prep_bands(w_origs, us, vs, need_crsk=True):
    return SpectralPrepBandsFn.apply(need_crsk, tuple(us), tuple(vs), *w_origs)


# ----------------------------------------------------------------------------- pointwise
ACT_NONE, ACT_APTX, ACT_LEAKY = 0, 1, 2


class PointwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, r_pre, r_post, mask, beta, gamma, cfg):
        act, mask_pre, mask_post, dmap, p_drop, seed = cfg
        Cc = h.shape[-1]
        rows = h.numel() // Cc
        d = PointwiseDesc(rows=rows, C=Cc, act=act, mask_pre=int(mask_pre), mask_post=int(mask_post), d1=dmap[0], d2=dmap[1],
                          d3=dmap[2], p_drop=float(p_drop), seed=seed)
        out = torch.empty_like(h)
        if h.dtype == BF16:
            check(L().mq_pointwise_fwd_h(C.byref(d), ptr(h, BF16), ptr(r_pre, BF16, True), ptr(r_post, BF16, True),
                                         ptr(mask, torch.uint8, True), fptr(beta, True), fptr(gamma, True), ptr(out, BF16), stream()),
                  "mq_pointwise_fwd_h")
        else:
            check(L().mq_pointwise_fwd(C.byref(d), fptr(h), fptr(r_pre, True), fptr(r_post, True), ptr(mask, torch.uint8, True),
                                       fptr(beta, True), fptr(gamma, True), fptr(out), stream()), "mq_pointwise_fwd")
        ctx.cfg = cfg
        ctx.save_for_backward(h, r_pre, mask, beta, gamma)
        ctx.has_post = r_post is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        h, r_pre, mask, beta, gamma = ctx.saved_tensors
        act, mask_pre, mask_post, dmap, p_drop, seed = ctx.cfg
        Cc = h.shape[-1]
        rows = h.numel() // Cc
        d = PointwiseDesc(rows=rows, C=Cc, act=act, mask_pre=int(mask_pre), mask_post=int(mask_post), d1=dmap[0], d2=dmap[1],
                          d3=dmap[2], p_drop=float(p_drop), seed=seed)
        dout = dout.contiguous()
        if dout.dtype != h.dtype:
            dout = cast_bf16(dout) if h.dtype == BF16 else cast_f32(dout)
        dt = torch.empty_like(h)
        dpost = torch.empty_like(h) if (ctx.has_post and ctx.needs_input_grad[2]) else None
        want_b = beta is not None and ctx.needs_input_grad[4]
        want_g = gamma is not None and ctx.needs_input_grad[5]
        dbeta = torch.empty_like(beta) if want_b else None
        dgamma = torch.empty_like(gamma) if want_g else None
        ws = _empty((2 * 1024,), h) if (want_b or want_g) else None
        if h.dtype == BF16:
            check(L().mq_pointwise_bwd_h(C.byref(d), ptr(dout, BF16), ptr(h, BF16), ptr(r_pre, BF16, True), ptr(mask, torch.uint8, True),
                                         fptr(beta, True), fptr(gamma, True), ptr(dt, BF16), ptr(dpost, BF16, True), fptr(dbeta, True),
                                         fptr(dgamma, True), fptr(ws, True), stream()), "mq_pointwise_bwd_h")
        else:
            check(L().mq_pointwise_bwd(C.byref(d), fptr(dout), fptr(h), fptr(r_pre, True), ptr(mask, torch.uint8, True),
                                       fptr(beta, True), fptr(gamma, True), fptr(dt), fptr(dpost, True), fptr(dbeta, True),
                                       fptr(dgamma, True), fptr(ws, True), stream()), "mq_pointwise_bwd")
        return dt, (dt if r_pre is not None else None), dpost, None, dbeta, dgamma, None


def pointwise_bwd_raw(dout, h, *, act, mask=None, mask_post=False, dmap=(1, 1, 1)):
    """dt of a parameter-free pointwise stage (no residuals, no dropout) - used by the fused conv epilogues' backward."""
    Cc = h.shape[-1]
    pd = PointwiseDesc(rows=h.numel() // Cc, C=Cc, act=act, mask_pre=0, mask_post=int(mask_post), d1=dmap[0], d2=dmap[1], d3=dmap[2],
                       p_drop=0.0, seed=0)
    dt = torch.empty_like(h)
    if h.dtype == BF16:
        check(L().mq_pointwise_bwd_h(C.byref(pd), ptr(dout, BF16), ptr(h, BF16), None, ptr(mask, torch.uint8, True), None, None,
                                     ptr(dt, BF16), None, None, None, None, stream()), "mq_pointwise_bwd_h")
    else:
        check(L().mq_pointwise_bwd(C.byref(pd), fptr(dout), fptr(h), None, ptr(mask, torch.uint8, True), None, None, fptr(dt), None,
                                   None, None, None, stream()), "mq_pointwise_bwd")
    return dt


def pointwise(h, *, act=ACT_NONE, r_pre=None, r_post=None, mask=None, mask_pre=False, mask_post=False, dmap=(1, 1, 1),
              beta=None, gamma=None, p_drop=0.0, training=False):
    """t = h (+ r_pre); [mask_pre]; act; dropout; (+ r_post); [mask_post].  ``dmap`` = (d1, d2, d3): the mask
    index of row m is (m // d1) * d2 + (m % d3)."""
    p = float(p_drop) if training else 0.0
    seed = next_seed() if p > 0 else 0
    if h.dtype == BF16:         # one storage type per launch
        r_pre = None if r_pre is None else to_bf16(r_pre)
        r_post = None if r_post is None else to_bf16(r_post)
    else:
        r_pre = None if r_pre is None else to_f32(r_pre)
        r_post = None if r_post is None else to_f32(r_post)
    return PointwiseFn.apply(h, r_pre, r_post, mask, beta, gamma, (act, bool(mask_pre), bool(mask_post), tuple(dmap), p, seed))


# ----------------------------------------------------------------------------- masks (no grad)
def sequence_mask(lengths: torch.Tensor, T: int) -> torch.Tensor:
    lengths = lengths.to(torch.int32).contiguous()
    B = lengths.numel()
    m = torch.empty((B, T), dtype=torch.uint8, device=lengths.device)
    check(L().mq_sequence_mask(ptr(lengths, torch.int32), B, T, ptr(m), stream()), "mq_sequence_mask")
    return m


def mask_pool(mask: torch.Tensor, stride: int, ceil_mode: bool) -> torch.Tensor:
    B, Tin = mask.shape
    Tout = -(-Tin // stride) if ceil_mode else Tin // stride
    out = torch.empty((B, Tout), dtype=torch.uint8, device=mask.device)
    check(L().mq_mask_pool(ptr(mask, torch.uint8), B, Tin, Tout, stride, ptr(out), stream()), "mq_mask_pool")
    return out


def mask_upsample2(mask: torch.Tensor) -> torch.Tensor:
    B, Tin = mask.shape
    out = torch.empty((B, 2 * Tin), dtype=torch.uint8, device=mask.device)
    check(L().mq_mask_upsample2(ptr(mask, torch.uint8), B, Tin, ptr(out), stream()), "mq_mask_upsample2")
    return out


# ----------------------------------------------------------------------------- UNet data movement
def _frame_mask(mask, N, T, what):
    """uint8 (N, T) mask of an op's OUTPUT frames (nonzero = padded), or None."""
    if mask is None:
        return None
    if mask.dtype != torch.uint8 or tuple(mask.shape) != (N, T):
        raise _lib.MqganHipError(f"{what}: output mask must be uint8 {(N, T)}, got {mask.dtype} {tuple(mask.shape)}")
    return mask.contiguous()


class AvgPoolT2Fn(torch.autograd.Function):
    """Time-only average pool / 2.  ``mask`` (N, T/2): padded output frames are written as zeros and pass no gradient -
    the entry masked_fill of the block that consumes the pooled map (preencoder.py:93), in the same pass."""

    @staticmethod
    def forward(ctx, x, mask=None):
        N, T, Fq, Cc = x.shape
        mask = _frame_mask(mask, N, T // 2, "avgpool_t2")
        y = _empty((N, T // 2, Fq, Cc), x, x.dtype)
        if x.dtype == BF16:
            check(L().mq_avgpool_t2_fwd_h(ptr(x, BF16), N, T, Fq, Cc, ptr(mask, torch.uint8, True), ptr(y, BF16), stream()),
                  "mq_avgpool_t2_fwd_h")
        else:
            check(L().mq_avgpool_t2_fwd(fptr(x), N, T, Fq, Cc, ptr(mask, torch.uint8, True), fptr(y), stream()), "mq_avgpool_t2_fwd")
        ctx.shape = (N, T, Fq, Cc)
        ctx.mask = mask
        return y

    @staticmethod
    def backward(ctx, dy):
        N, T, Fq, Cc = ctx.shape
        dy = dy.contiguous()
        dx = _empty(ctx.shape, dy, dy.dtype)
        m = ptr(ctx.mask, torch.uint8, True)
        if dy.dtype == BF16:
            check(L().mq_avgpool_t2_bwd_h(ptr(dy, BF16), N, T, Fq, Cc, m, ptr(dx, BF16), stream()), "mq_avgpool_t2_bwd_h")
        else:
            check(L().mq_avgpool_t2_bwd(fptr(dy), N, T, Fq, Cc, m, fptr(dx), stream()), "mq_avgpool_t2_bwd")
        return dx, None


class UpCatFn(torch.autograd.Function):
    """cat(nearest-upsample x2 in time of x, skip) along channels.  ``mask`` (N, 2T): as in AvgPoolT2Fn."""

    @staticmethod
    def forward(ctx, x, skip, mask=None):
        N, T, Fq, C1 = x.shape
        C2 = skip.shape[-1]
        if tuple(skip.shape[:3]) != (N, 2 * T, Fq):
            raise _lib.MqganHipError(f"upcat: skip {tuple(skip.shape)} does not match upsampled {(N, 2 * T, Fq)}")
        if x.dtype != skip.dtype:
            raise _lib.MqganHipError(f"upcat: dtype mismatch {x.dtype} vs {skip.dtype}")
        mask = _frame_mask(mask, N, 2 * T, "upcat")
        m = ptr(mask, torch.uint8, True)
        out = _empty((N, 2 * T, Fq, C1 + C2), x, x.dtype)
        if x.dtype == BF16:
            check(L().mq_upcat_fwd_h(ptr(x, BF16), ptr(skip, BF16), N, T, Fq, C1, C2, m, ptr(out, BF16), stream()), "mq_upcat_fwd_h")
        else:
            check(L().mq_upcat_fwd(fptr(x), fptr(skip), N, T, Fq, C1, C2, m, fptr(out), stream()), "mq_upcat_fwd")
        ctx.dims = (N, T, Fq, C1, C2)
        ctx.mask = mask
        return out

    @staticmethod
    def backward(ctx, dout):
        N, T, Fq, C1, C2 = ctx.dims
        dout = dout.contiguous()
        dx = _empty((N, T, Fq, C1), dout, dout.dtype)
        ds = _empty((N, 2 * T, Fq, C2), dout, dout.dtype)
        m = ptr(ctx.mask, torch.uint8, True)
        if dout.dtype == BF16:
            check(L().mq_upcat_bwd_h(ptr(dout, BF16), N, T, Fq, C1, C2, m, ptr(dx, BF16), ptr(ds, BF16), stream()), "mq_upcat_bwd_h")
        else:
            check(L().mq_upcat_bwd(fptr(dout), N, T, Fq, C1, C2, m, fptr(dx), fptr(ds), stream()), "mq_upcat_bwd")
        return dx, ds, None


def mask_fusable(*channels):
    """The pool / upsample-concat kernels take the consumer's frame mask only on their vectorised path."""
    return all(c % 8 == 0 for c in channels)


class BandSplitFn(torch.autograd.Function):
    """(B, T, F) -> (n_bins*B, F/n_bins, T, 1): the (H = mel, W = time) single-channel image the
    discriminators convolve (discriminators.py:211, 298).  n_bins = 1 is a plain transpose."""

    @staticmethod
    def forward(ctx, x, n_bins):
        B, T, Fq = x.shape
        out = _empty((n_bins * B, Fq // n_bins, T, 1), x)
        check(L().mq_band_split_fwd(fptr(x), B, T, Fq, n_bins, fptr(out), stream()), "mq_band_split_fwd")
        ctx.dims = (B, T, Fq, n_bins)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, Fq, n_bins = ctx.dims
        dx = _empty((B, T, Fq), dout)
        check(L().mq_band_split_bwd(fptr(dout.contiguous()), B, T, Fq, n_bins, fptr(dx), stream()), "mq_band_split_bwd")
        return dx, None


class Concat2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        C1, C2 = a.shape[-1], b.shape[-1]
        rows = a.numel() // C1
        y = _empty(tuple(a.shape[:-1]) + (C1 + C2,), a)
        check(L().mq_concat2(fptr(a), fptr(b), rows, C1, C2, fptr(y), stream()), "mq_concat2")
        ctx.c = (C1, C2)
        return y

    @staticmethod
    def backward(ctx, dy):  # only used on detached inputs in this model; provided for completeness
        C1, C2 = ctx.c
        return dy[..., :C1].contiguous(), dy[..., C1:].contiguous()


class AddFn(torch.autograd.Function):
    """out = alpha*a + beta*b on the device kernel (keeps elementwise arithmetic out of ATen)."""

    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        out = torch.empty_like(a)
        check(L().mq_axpby(fptr(a), fptr(b), alpha, beta, a.numel(), fptr(out), stream()), "mq_axpby")
        ctx.ab = (alpha, beta)
        return out

    @staticmethod
    def backward(ctx, d):
        alpha, beta = ctx.ab
        d = d.contiguous()
        return scale(d, alpha), scale(d, beta), None, None


def scale(a, alpha):
    if alpha == 1.0:
        return a
    out = torch.empty_like(a)
    check(L().mq_axpby(fptr(a), None, float(alpha), 0.0, a.numel(), fptr(out), stream()), "mq_axpby")
    return out


def add(a, b, alpha=1.0, beta=1.0):
    return AddFn.apply(a, b, float(alpha), float(beta))


# ----------------------------------------------------------------------------- small activations on (B, C) tensors
class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        check(L().mq_relu_fwd(fptr(x), x.numel(), fptr(y), stream()), "mq_relu_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        check(L().mq_relu_bwd(fptr(dy.contiguous()), fptr(y), y.numel(), fptr(dx), stream()), "mq_relu_bwd")
        return dx


class SigmoidAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        check(L().mq_sigmoid_add(fptr(a), fptr(b, True), a.numel(), fptr(out), stream()), "mq_sigmoid_add")
        ctx.save_for_backward(out)
        ctx.has_b = b is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        da = torch.empty_like(out)
        check(L().mq_sigmoid_add_bwd(fptr(dout.contiguous()), fptr(out), out.numel(), fptr(da), stream()), "mq_sigmoid_add_bwd")
        return da, (da if ctx.has_b else None)


# ----------------------------------------------------------------------------- CBAM
class CamPoolFn(torch.autograd.Function):
    """x (B, T, C), mask (B, T) -> (max over ALL t, mean over VALID t), each (B, C)."""

    @staticmethod
    def forward(ctx, x, mask):
        B, T, Cc = x.shape
        mx, av = _empty((B, Cc), x), _empty((B, Cc), x)
        am = _empty((B, Cc), x, torch.int32)
        check(L().mq_cam_pool(fptr(x), ptr(mask, torch.uint8), B, T, Cc, fptr(mx), ptr(am), fptr(av), stream()), "mq_cam_pool")
        ctx.save_for_backward(am, mask)
        ctx.dims = (B, T, Cc)
        ctx.mark_non_differentiable(am)
        return mx, av

    @staticmethod
    def backward(ctx, dmx, dav):
        am, mask = ctx.saved_tensors
        B, T, Cc = ctx.dims
        dx = torch.zeros((B, T, Cc), dtype=F32, device=dmx.device)
        check(L().mq_cam_pool_bwd(fptr(dmx.contiguous()), fptr(dav.contiguous()), ptr(am), ptr(mask, torch.uint8), B, T, Cc,
                                  fptr(dx), stream()), "mq_cam_pool_bwd")
        return dx, None


class CbamApplyFn(torch.autograd.Function):
    """out = sigmoid(conv_k(pool_c(gate*x))) * gate * x + x   (attentions.py:310-365, 411)."""

    @staticmethod
    def forward(ctx, x, gate, w7):
        B, T, Cc = x.shape
        ks = w7.shape[-1]
        feat = _empty((B, T, 2), x)
        amc = _empty((B, T), x, torch.int32)
        check(L().mq_sam_pool(fptr(x), fptr(gate), B, T, Cc, fptr(feat), ptr(amc), stream()), "mq_sam_pool")
        s = _empty((B, T), x)
        out = torch.empty_like(x)
        check(L().mq_sam_apply(fptr(x), fptr(gate), fptr(feat), fptr(w7), ks, B, T, Cc, fptr(s), fptr(out), stream()), "mq_sam_apply")
        ctx.save_for_backward(x, gate, feat, amc, s, w7)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, gate, feat, amc, s, w7 = ctx.saved_tensors
        B, T, Cc = x.shape
        ks = w7.shape[-1]
        dx, dgate, dw7 = torch.empty_like(x), torch.empty_like(gate), torch.empty_like(w7)
        ws = _ws_bytes(L().mq_cbam_bwd_workspace(B, T, Cc, ks), x)
        check(L().mq_cbam_bwd(fptr(dout.contiguous()), fptr(x), fptr(gate), fptr(feat), ptr(amc), fptr(s), fptr(w7), ks, B, T, Cc,
                              fptr(dx), fptr(dgate), fptr(dw7), ptr(ws), stream()), "mq_cbam_bwd")
        return dx, dgate, dw7


# ----------------------------------------------------------------------------- masked SE
class SeSqueezeSkipFn(torch.autograd.Function):
    """Masked mean over (H, W) for the squeeze-excite block, for the usual case that x goes on to SeScaleFn: returns (x, squeeze(x)) and, in backward, ADDS the squeeze's
    broadcast gradient into the gradient that arrived for x (mq_se_squeeze_bwd accumulates) instead of handing autograd a second
    full-size tensor to sum - one fill and one three-stream add over the discriminator's widest feature map less per pass, same
    two-operand additions (bit-identical).  The incoming gradient is updated in place: it is the tensor SeScaleFn.backward (or
    autograd's own accumulation, with several consumers) produced for this edge alone."""

    @staticmethod
    def forward(ctx, x, mask):
        B, H, W, Cc = x.shape
        sq = _empty((B, Cc), x)
        check(L().mq_se_squeeze(fptr(x), ptr(mask, torch.uint8), B, H, W, Cc, fptr(sq), stream()), "mq_se_squeeze")
        ctx.save_for_backward(mask)
        ctx.dims = (B, H, W, Cc)
        return x, sq

    @staticmethod
    def backward(ctx, dx, dsq):
        (mask,) = ctx.saved_tensors
        B, H, W, Cc = ctx.dims
        if dsq is None:
            return dx, None
        if dx is None:
            dx = torch.zeros(ctx.dims, dtype=F32, device=dsq.device)
        elif not dx.is_contiguous() or dx.dtype != F32:
            dx = dx.to(F32).contiguous()
        check(L().mq_se_squeeze_bwd(fptr(dsq.contiguous()), ptr(mask, torch.uint8), B, H, W, Cc, fptr(dx), stream()),
              "mq_se_squeeze_bwd")
        return dx, None


class SeScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, excite):
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        check(L().mq_se_scale_fwd(fptr(x), fptr(excite), B, H * W, Cc, fptr(y), stream()), "mq_se_scale_fwd")
        ctx.save_for_backward(x, excite)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, excite = ctx.saved_tensors
        B, H, W, Cc = x.shape
        dx, de = torch.empty_like(x), torch.empty_like(excite)
        check(L().mq_se_scale_bwd(fptr(dy.contiguous()), fptr(x), fptr(excite), B, H * W, Cc, fptr(dx), fptr(de), stream()),
              "mq_se_scale_bwd")
        return dx, de


# ----------------------------------------------------------------------------- fused ConvBlock2D
class ConvBlock2DFn(torch.autograd.Function):
    """x (B, T, C0) -> y (B, T, C0); effective weights wdw (25), bdw (1), w1 (C0), b1 (C0), w2 (C0), b2 (1)."""

    @staticmethod
    def forward(ctx, x, mask, wdw, bdw, w1, b1, w2, b2, p_drop, seed):
        B, T, C0 = x.shape
        w = Cb2dWeights(wdw=wdw.data_ptr(), bdw=bdw.data_ptr(), w1=w1.data_ptr(), b1=b1.data_ptr(), w2=w2.data_ptr(),
                        b2=b2.data_ptr())
        for t in (wdw, bdw, w1, b1, w2, b2):
            fptr(t)
        y = torch.empty_like(x)
        v = torch.empty_like(x)
        # dropout decisions: drawn once here, one bit per (pixel, hidden unit), read back by the backward kernel
        bits = _ws_bytes(L().mq_convblock2d_keep_bytes(B, T, C0), x) if p_drop > 0 else None
        check(L().mq_convblock2d_fwd(C.byref(w), fptr(x), ptr(mask, torch.uint8), B, T, C0, float(p_drop), seed, fptr(y), fptr(v),
                                     ptr(bits, torch.uint8, True), stream()), "mq_convblock2d_fwd")
        ctx.save_for_backward(x, v, mask, wdw, bdw, w1, b1, w2, b2, bits)
        ctx.cfg = (float(p_drop), seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, v, mask, wdw, bdw, w1, b1, w2, b2, bits = ctx.saved_tensors
        B, T, C0 = x.shape
        p_drop, seed = ctx.cfg
        w = Cb2dWeights(wdw=wdw.data_ptr(), bdw=bdw.data_ptr(), w1=w1.data_ptr(), b1=b1.data_ptr(), w2=w2.data_ptr(),
                        b2=b2.data_ptr())
        dx = torch.empty_like(x)
        g = [torch.empty_like(t) for t in (wdw, bdw, w1, b1, w2, b2)]
        ws = _ws_bytes(L().mq_convblock2d_bwd_workspace(B, T, C0), x)
        check(L().mq_convblock2d_bwd(C.byref(w), fptr(x), fptr(v), fptr(dy.contiguous()), ptr(mask, torch.uint8), B, T, C0, p_drop,
                                     ptr(bits, torch.uint8, True), fptr(dx), fptr(g[0]), fptr(g[1]), fptr(g[2]), fptr(g[3]), fptr(g[4]), fptr(g[5]), ptr(ws),
                                     stream()), "mq_convblock2d_bwd")
        return (dx, None) + tuple(g) + (None, None)


# ----------------------------------------------------------------------------- FSQ / VQ
class FsqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, desc):
        d = desc.d
        n = z.numel() // d
        codes = torch.empty_like(z)
        idx = torch.empty(z.shape[:-1], dtype=torch.int32, device=z.device)
        check(L().mq_fsq_forward(C.byref(desc), fptr(z), n, fptr(codes), ptr(idx), stream()), "mq_fsq_forward")
        ctx.save_for_backward(z)
        ctx.desc = desc
        ctx.mark_non_differentiable(idx)
        return codes, idx

    @staticmethod
    def backward(ctx, dcodes, _):
        (z,) = ctx.saved_tensors
        dz = torch.empty_like(z)
        n = z.numel() // ctx.desc.d
        check(L().mq_fsq_backward(C.byref(ctx.desc), fptr(z), fptr(dcodes.contiguous()), n, fptr(dz), stream()), "mq_fsq_backward")
        return dz, None


def fsq_decode(idx: torch.Tensor, desc) -> torch.Tensor:
    idx = idx.to(torch.int64).contiguous()
    codes = torch.empty(tuple(idx.shape) + (desc.d,), dtype=F32, device=idx.device)
    check(L().mq_fsq_decode(C.byref(desc), ptr(idx, torch.int64), idx.numel(), fptr(codes), stream()), "mq_fsq_decode")
    return codes


def l2_codebook_norms(codebook: torch.Tensor) -> torch.Tensor:
    k, d = codebook.shape
    cn = _empty((k,), codebook)
    check(L().mq_l2_codebook_norms(fptr(codebook), k, d, fptr(cn), stream()), "mq_l2_codebook_norms")
    return cn


def l2_argmin(x: torch.Tensor, codebook: torch.Tensor, cnorm: Optional[torch.Tensor] = None, return_dist=False,
              workspace: Optional[torch.Tensor] = None):
    n, d = x.shape
    k = codebook.shape[0]
    if cnorm is None:
        cnorm = l2_codebook_norms(codebook)
    idx = torch.empty((n,), dtype=torch.int32, device=x.device)
    dist = _empty((n,), x) if return_dist else None
    if workspace is None:
        workspace = _ws_bytes(L().mq_l2_argmin_workspace(n), x)
    check(L().mq_l2_argmin(fptr(x), fptr(codebook), fptr(cnorm), n, k, d, ptr(idx), fptr(dist, True), ptr(workspace), stream()),
          "mq_l2_argmin")
    return (idx, dist) if return_dist else idx


def l2_screen_supported(n, k, d) -> bool:
    return bool(L().mq_l2_screen_supported(n, k, d))


def l2_screen_prepare(codebook: torch.Tensor) -> torch.Tensor:
    """Per-codebook data of the screened search (code norms, bf16 copy, largest norm): once per codebook version."""
    k, d = codebook.shape
    blob = _ws_bytes(L().mq_l2_screen_codebook_bytes(k, d), codebook)
    check(L().mq_l2_screen_prepare(fptr(codebook), k, d, ptr(blob), stream()), "mq_l2_screen_prepare")
    return blob


def l2_argmin_screened(x: torch.Tensor, codebook: torch.Tensor, prepared: Optional[torch.Tensor] = None, return_dist=False,
                       workspace: Optional[torch.Tensor] = None):
    """mq_l2_argmin's results (bit for bit) from bf16 screening + exact f32 re-rank (csrc/vq_screen.hip)."""
    n, d = x.shape
    k = codebook.shape[0]
    if prepared is None:
        prepared = l2_screen_prepare(codebook)
    idx = torch.empty((n,), dtype=torch.int32, device=x.device)
    dist = _empty((n,), x) if return_dist else None
    nbytes = L().mq_l2_screen_workspace(n, k, d)
    if workspace is None:
        workspace = _ws_bytes(nbytes, x)
    check(L().mq_l2_argmin_screened(fptr(x), fptr(codebook), ptr(prepared), n, k, d, ptr(idx), fptr(dist, True), ptr(workspace),
                                    workspace.numel() * workspace.element_size(), stream()), "mq_l2_argmin_screened")
    return (idx, dist) if return_dist else idx


def l2_screen_stats(workspace: torch.Tensor, n, k, d):
    """(list entries appended, list entries available, rows searched exhaustively) of the last screened search on
    ``workspace``; synchronises the stream."""
    import ctypes as _C
    a, sl, ov = _C.c_uint64(0), _C.c_uint64(0), _C.c_uint64(0)
    check(L().mq_l2_screen_stats(ptr(workspace), n, k, d, _C.byref(a), _C.byref(sl), _C.byref(ov), stream()), "mq_l2_screen_stats")
    return a.value, sl.value, ov.value


def codebook_ema_update(x, idx, cluster_size, embed_sum, codebook, decay=0.99, eps=1e-5):
    """In-place EMA update of (cluster_size, embed_sum, codebook); deterministic, bit-exact against oracle/vq_ref.c."""
    n, d = x.shape
    k = codebook.shape[0]
    nbytes = L().mq_codebook_ema_workspace(n, k, d)
    ws = _ws_bytes(nbytes, x)
    check(L().mq_codebook_ema_update(fptr(x), ptr(idx, torch.int32), n, k, d, decay, eps, fptr(cluster_size), fptr(embed_sum),
                                     fptr(codebook), ptr(ws), nbytes, stream()), "mq_codebook_ema_update")


# ----------------------------------------------------------------------------- losses
# Every loss is computed in two halves so that the normalisers can be made global under data parallelism: the *_sums
# kernels leave numerators and counts in a small device buffer, ``comm.all_reduce_sum`` (identity for one process) sums it
# over the ranks, and the *_final / backward kernels work from the reduced buffer (see include/mqgan_hip.h).
def _reduce(comm, buf):
    if comm is not None and comm.active:
        comm.all_reduce_sum(buf)
    return buf


class SumsBatch:
    """The numerators / counts of SEVERAL losses in one device buffer, so that they cross the data-parallel ranks in ONE
    all-reduce: ``take(n)`` hands out the next n floats (16-byte aligned) for a *_sums kernel to fill, ``reduce()`` sums the
    used part over the ranks once, and the *_final / backward kernels of every loss then read their slice.  The generator step
    uses one batch for its two mel losses, the nine LSGAN generator losses and (with feature matching) every masked_mae:
    one small collective instead of eleven or more (each of them a compute-stream <-> RCCL-stream round trip)."""

    def __init__(self, like: torch.Tensor, comm=None, capacity: int = 4096):
        self.buf = _empty((capacity,), like)
        self.used = 0
        self.comm = comm
        self.reduced = False

    def take(self, n: int, rows: int = 1) -> torch.Tensor:
        if self.reduced:
            raise _lib.MqganHipError("SumsBatch: take() after reduce()")
        total = n * rows
        if self.used + total > self.buf.numel():
            raise _lib.MqganHipError(f"SumsBatch: capacity {self.buf.numel()} floats exceeded")
        out = self.buf[self.used:self.used + total]
        self.used += (total + 3) // 4 * 4
        return out if rows == 1 else out.view(rows, n)

    def reduce(self):
        if self.reduced:
            raise _lib.MqganHipError("SumsBatch: reduce() called twice")
        self.reduced = True
        if self.used:
            _reduce(self.comm, self.buf[:self.used])


def _sums_ready(prepared):
    """A loss Function got its sums from a SumsBatch: they must have crossed the ranks before the final kernel reads them."""
    batch, sums = prepared
    if not batch.reduced:
        raise _lib.MqganHipError("loss from a SumsBatch evaluated before SumsBatch.reduce()")
    return sums


def mel_loss_sums(x, y, lengths, batch: SumsBatch):
    """First half of MelLossFn into a SumsBatch slice; pass the result as ``prepared``."""
    B, T, Cc = x.shape
    sums = batch.take(Cc + 1)
    ws = _empty((64 * Cc,), x)
    check(L().mq_mel_loss_sums(fptr(x), fptr(y), ptr(lengths, torch.int32), B, T, Cc, fptr(sums), fptr(ws), stream()),
          "mq_mel_loss_sums")
    return (batch, sums)


def lsgan_g_sums(fake, pad, mode, batch: SumsBatch):
    B, H, W = fake.shape
    sums2 = batch.take(2)
    check(L().mq_lsgan_g_sums(fptr(fake), ptr(pad, torch.uint8, True), mode, B, H, W, fptr(sums2), stream()), "mq_lsgan_g_sums")
    return (batch, sums2)


def masked_mae_sums(pred, target, padded, batch: SumsBatch, bands: int = 1):
    """bands == 1: MaskedMaeFn's sums (2,); else MaskedMaeBandsFn's (bands, 2) for the row blocks of a multi-band feature map."""
    GB, H, W, Cc = pred.shape
    Bn = GB // bands
    ws = _empty((512,), pred)
    sums2 = batch.take(2, rows=bands) if bands > 1 else batch.take(2)
    for j in range(bands):
        check(L().mq_masked_mae_sums(fptr(pred[j * Bn:(j + 1) * Bn]), fptr(target[j * Bn:(j + 1) * Bn]), ptr(padded, torch.uint8),
                                     Bn, H, W, Cc, fptr(sums2[j] if bands > 1 else sums2), fptr(ws), stream()), "mq_masked_mae_sums")
    return (batch, sums2)


class MelLossFn(torch.autograd.Function):
    """MaskedMelLoss('mse') for group sizes g1 and g2 in one pass: returns w1*loss(g1) + w2*loss(g2)."""

    @staticmethod
    def forward(ctx, x, y, lengths, g1, g2, w1, w2, comm, prepared=None):
        B, T, Cc = x.shape
        if prepared is not None:
            sums = _sums_ready(prepared)
        else:
            sums = _empty((Cc + 1,), x)
            ws = _empty((64 * Cc,), x)
            check(L().mq_mel_loss_sums(fptr(x), fptr(y), ptr(lengths, torch.int32), B, T, Cc, fptr(sums), fptr(ws), stream()),
                  "mq_mel_loss_sums")
            _reduce(comm, sums)
        out2 = _empty((2,), x)
        check(L().mq_mel_loss_final(fptr(sums), Cc, g1, g2, fptr(out2), stream()), "mq_mel_loss_final")
        ctx.save_for_backward(x, y, lengths, sums)
        ctx.cfg = (g1, g2, w1, w2)
        out = _empty((1,), x)
        check(L().mq_axpby(fptr(out2[0:1]), fptr(out2[1:2]), float(w1), float(w2), 1, fptr(out), stream()), "mq_axpby")
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        x, y, lengths, sums = ctx.saved_tensors
        g1, g2, w1, w2 = ctx.cfg
        B, T, Cc = x.shape
        dx = torch.empty_like(x)
        check(L().mq_mel_loss_bwd(fptr(x), fptr(y), ptr(lengths, torch.int32), B, T, Cc, g1, g2, w1, w2, fptr(sums),
                                  fptr(gout.contiguous().view(1)), fptr(dx), stream()), "mq_mel_loss_bwd")
        return dx, None, None, None, None, None, None, None, None


def lsgan_d_prepare(pairs, state, decay, comm=None):
    """First half of n consecutive LSGANLoss.discriminator_loss calls that share one EMA state (train.py:385-396):
    ``pairs`` = [(real (B,H,W), fake, real_pad, fake_pad, mask_mode)]; returns (sums (n,6), ema (n,2), lsums (n,2)).
    Two small all-reduces for the whole set (sums before the EMA update, LeCam sums after it)."""
    n = len(pairs)
    like = pairs[0][0]
    sums, ema, lsums = _empty((n, 6), like), _empty((n, 2), like), _empty((n, 2), like)
    for i, (r, f, rp, fp, mode) in enumerate(pairs):
        B, H, W = r.shape
        check(L().mq_lsgan_d_sums(fptr(r), fptr(f), ptr(rp, torch.uint8, True), ptr(fp, torch.uint8, True), mode, B, H, W, fptr(sums[i]),
                                  stream()), "mq_lsgan_d_sums")
    _reduce(comm, sums)
    check(L().mq_lsgan_ema(fptr(sums), n, float(decay), fptr(state), fptr(ema), stream()), "mq_lsgan_ema")
    for i, (r, f, rp, fp, mode) in enumerate(pairs):
        B, H, W = r.shape
        check(L().mq_lsgan_d_lecam(fptr(r), fptr(f), ptr(rp, torch.uint8, True), ptr(fp, torch.uint8, True), mode, B, H, W, fptr(ema[i]),
                                   fptr(lsums[i]), stream()), "mq_lsgan_d_lecam")
    _reduce(comm, lsums)
    return sums, ema, lsums


class LsganDFn(torch.autograd.Function):
    """Second half: the loss of one (real, fake) pair from its prepared sums."""

    @staticmethod
    def forward(ctx, real, fake, rpad, fpad, mode, sums6, ema2, lsums2):
        out = _empty((1,), real)
        check(L().mq_lsgan_d_final(fptr(sums6), fptr(lsums2), fptr(out), stream()), "mq_lsgan_d_final")
        ctx.save_for_backward(real, fake, rpad, fpad, sums6, ema2)
        ctx.mode = mode
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        real, fake, rpad, fpad, sums6, ema2 = ctx.saved_tensors
        B, H, W = real.shape
        dr, df = torch.empty_like(real), torch.empty_like(fake)
        check(L().mq_lsgan_d_bwd(fptr(real), fptr(fake), ptr(rpad, torch.uint8, True), ptr(fpad, torch.uint8, True), ctx.mode, B, H, W,
                                 fptr(sums6), fptr(ema2), fptr(gout.contiguous().view(1)), fptr(dr), fptr(df), stream()), "mq_lsgan_d_bwd")
        return dr, df, None, None, None, None, None, None


class LsganGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fake, pad, mode, comm, prepared=None):
        B, H, W = fake.shape
        out = _empty((1,), fake)
        if prepared is not None:
            sums2 = _sums_ready(prepared)
        else:
            sums2 = _empty((2,), fake)
            check(L().mq_lsgan_g_sums(fptr(fake), ptr(pad, torch.uint8, True), mode, B, H, W, fptr(sums2), stream()), "mq_lsgan_g_sums")
            _reduce(comm, sums2)
        check(L().mq_ratio_final(fptr(sums2), 0.0, fptr(out), stream()), "mq_ratio_final")
        ctx.save_for_backward(fake, pad, sums2)
        ctx.mode = mode
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        fake, pad, sums2 = ctx.saved_tensors
        B, H, W = fake.shape
        df = torch.empty_like(fake)
        check(L().mq_lsgan_g_bwd(fptr(fake), ptr(pad, torch.uint8, True), ctx.mode, B, H, W, fptr(sums2), fptr(gout.contiguous().view(1)),
                                 fptr(df), stream()), "mq_lsgan_g_bwd")
        return df, None, None, None, None


class MaskedMaeFn(torch.autograd.Function):
    """``leaky_slope``: pred is the output of a ``pregated`` LeakyReLU layer (conv_leaky): the gradient handed back is the one
    with respect to that layer's pre-activation."""

    @staticmethod
    def forward(ctx, pred, target, padded, comm, leaky_slope=None, prepared=None):
        B, H, W, Cc = pred.shape
        out = _empty((1,), pred)
        if prepared is not None:
            sums2 = _sums_ready(prepared)
        else:
            sums2, ws = _empty((2,), pred), _empty((512,), pred)
            check(L().mq_masked_mae_sums(fptr(pred), fptr(target), ptr(padded, torch.uint8), B, H, W, Cc, fptr(sums2), fptr(ws), stream()),
                  "mq_masked_mae_sums")
            _reduce(comm, sums2)
        check(L().mq_ratio_final(fptr(sums2), 1e-8, fptr(out), stream()), "mq_ratio_final")
        ctx.save_for_backward(pred, target, padded, sums2)
        ctx.leaky_slope = 1.0 if leaky_slope is None else float(leaky_slope)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        pred, target, padded, sums2 = ctx.saved_tensors
        B, H, W, Cc = pred.shape
        dp = torch.empty_like(pred)
        check(L().mq_masked_mae_bwd(fptr(pred), fptr(target), ptr(padded, torch.uint8), B, H, W, Cc, fptr(sums2),
                                    fptr(gout.contiguous().view(1)), ctx.leaky_slope, fptr(dp), stream()), "mq_masked_mae_bwd")
        return dp, None, None, None, None, None


class MaskedMaeBandsFn(torch.autograd.Function):
    """masked_mae of every band of a multi-band feature map in one autograd node: pred / target (G*Bn, H, W, C) with band j
    in rows [j*Bn, (j+1)*Bn), one padded mask (Bn, W) for all bands (the reference uses bin 0's) -> (G,) losses.
    The backward writes each band's slice of ONE gradient tensor (slicing the feature per band instead costs a zero-filled
    full-size tensor and an accumulation per band); the G normalisers travel in one all-reduce."""

    @staticmethod
    def forward(ctx, pred, target, padded, G, comm, leaky_slope=None, prepared=None):
        GB, H, W, Cc = pred.shape
        Bn = GB // G
        out = _empty((G,), pred)
        if prepared is not None:
            sums2 = _sums_ready(prepared)
        else:
            sums2, ws = _empty((G, 2), pred), _empty((512,), pred)
            for j in range(G):
                check(L().mq_masked_mae_sums(fptr(pred[j * Bn:(j + 1) * Bn]), fptr(target[j * Bn:(j + 1) * Bn]), ptr(padded, torch.uint8),
                                             Bn, H, W, Cc, fptr(sums2[j]), fptr(ws), stream()), "mq_masked_mae_sums")
            _reduce(comm, sums2)
        for j in range(G):
            check(L().mq_ratio_final(fptr(sums2[j]), 1e-8, fptr(out[j:j + 1]), stream()), "mq_ratio_final")
        ctx.save_for_backward(pred, target, padded, sums2)
        ctx.G = G
        ctx.leaky_slope = 1.0 if leaky_slope is None else float(leaky_slope)
        return out

    @staticmethod
    def backward(ctx, gout):
        pred, target, padded, sums2 = ctx.saved_tensors
        GB, H, W, Cc = pred.shape
        G = ctx.G
        Bn = GB // G
        gout = gout.contiguous()
        dp = torch.empty_like(pred)
        for j in range(G):
            check(L().mq_masked_mae_bwd(fptr(pred[j * Bn:(j + 1) * Bn]), fptr(target[j * Bn:(j + 1) * Bn]), ptr(padded, torch.uint8),
                                        Bn, H, W, Cc, fptr(sums2[j]), fptr(gout[j:j + 1]), ctx.leaky_slope,
                                        fptr(dp[j * Bn:(j + 1) * Bn]), stream()), "mq_masked_mae_bwd")
        return dp, None, None, None, None, None, None


# ----------------------------------------------------------------------------- optimiser
def sumsq(flat: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = _empty((1,), flat) if out is None else out
    ws = _empty((1024,), flat)
    check(L().mq_sumsq(fptr(flat), flat.numel(), fptr(out), fptr(ws), stream()), "mq_sumsq")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, sumsq_t=None, max_norm=0.0, grad_scale=1.0):
    check(L().mq_adam_step(fptr(param), fptr(grad), fptr(exp_avg), fptr(exp_avg_sq), param.numel(), float(lr), float(beta1),
                           float(beta2), float(eps), int(step), fptr(sumsq_t, True), float(max_norm), float(grad_scale),
                           stream()), "mq_adam_step")
