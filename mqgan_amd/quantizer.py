"""FSQ with the reference's interface (quantizer.py:50-272, the configuration the PreEncoder
uses: dim == len(levels), one codebook, indices returned) on the HIP kernels, plus the generic
L2 nearest-neighbour quantiser the north-star benchmarks beside it.

The index search is threshold based: the reference pipeline
    level(z) = rint(f32(f32(tanh(z + shift)) * half_l) - offset) + L // 2
is monotone in z, so it is fully described by L - 1 float thresholds per dimension.  They are
found here, on the host, by bisection over float bit patterns with exactly that arithmetic, and the
kernel only counts ``z >= theta``.

Which ``tanh``?  The reference calls ``torch.tanh`` on an f32 tensor (quantizer.py:113), i.e. whatever
f32 tanh the device / host library provides - not one function: it differs by an ulp between libraries.
  * ``tanh_impl="exact"`` (default): tanh in double, rounded once to f32 (the correctly rounded value).
    Platform independent; this is the definition oracle/vq_ref.c restates and the kernel is bit-exact to.
    Against torch-CPU's vectorised f32 tanh on the build host, 2 of the 19 thresholds of levels [8,5,5,5]
    sit one ulp away (tests/test_oracle_vq.py lists them): an index differs only when z IS that float,
    about 3e-8 per row for z ~ N(0, 2^2) (measured: 1-2 rows in 2^26).
  * ``tanh_impl="torch"`` (opt-in, or MQGAN_FSQ_TANH=torch): the thresholds are bisected with THIS host's
    ``torch.tanh`` on f32 tensors, after checking that it is monotone in a +-64-ulp window around every
    threshold: indices then equal torch-CPU's on this host by construction, for every float.
"""
import math
from typing import List

import numpy as np
import torch
import torch.nn as nn

from . import ops
from ._lib import FsqDesc


def _f2o(x: np.float32) -> int:
    i = int(np.float32(x).view(np.int32))
    return i if i >= 0 else -(i & 0x7FFFFFFF)


def _o2f(o: int) -> np.float32:
    i = o if o >= 0 else ((-o) | 0x80000000)
    return np.uint32(i).view(np.float32)


def _level(z: np.float32, half_l: np.float32, offset: np.float32, shift: np.float32, half_width: int) -> int:
    t = np.float32(z) + shift                                   # f32 add
    th = np.float32(math.tanh(float(t)))                        # tanh in double, one rounding to f32
    b = np.float32(th * half_l) - offset                        # f32 mul, f32 sub (separately rounded)
    return int(np.rint(b)) + half_width                         # half-to-even


def _level_torch(z: np.float32, half_l, offset, shift, half_width: int) -> int:
    """The same pipeline on this host's torch (f32 tensors, torch.tanh): 16 copies so that the vectorised path runs."""
    return int(_levels_torch(np.full(16, z, dtype=np.float32), half_l, offset, shift, half_width)[0])


def _levels_torch(zs: np.ndarray, half_l, offset, shift, half_width: int) -> np.ndarray:
    z = torch.from_numpy(np.ascontiguousarray(zs, dtype=np.float32))
    b = torch.tanh(z + float(shift)) * float(half_l) - float(offset)        # quantizer.py:111-114: f32 throughout
    return (torch.round(b).to(torch.int32) + half_width).numpy()


def fsq_constants(levels: List[int]):
    """Same torch ops / dtypes as quantizer.py:68-72,111-113,132, so the constants are bit-identical."""
    lv = torch.tensor(levels, dtype=torch.int32)
    half_l = (lv - 1) * (1 + 1e-3) / 2
    offset = torch.where(lv % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    half_width = lv // 2
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1])), dim=0, dtype=torch.int32)
    return lv, half_l, offset, shift, half_width, basis


def make_fsq_desc(levels: List[int], tanh_impl: str = "exact") -> FsqDesc:
    if not 1 <= len(levels) <= 8 or any(not 2 <= l <= 9 for l in levels):
        raise ValueError("FSQ: 1..8 dimensions with 2..9 levels each are supported")
    if tanh_impl not in ("exact", "torch"):
        raise ValueError("FSQ: tanh_impl must be 'exact' or 'torch'")
    level = _level if tanh_impl == "exact" else _level_torch
    lv, half_l, offset, shift, half_width, basis = fsq_constants(levels)
    d = FsqDesc()
    d.d = len(levels)
    for j, L in enumerate(levels):
        hl, of, sh = np.float32(half_l[j].item()), np.float32(offset[j].item()), np.float32(shift[j].item())
        hw = int(half_width[j])
        d.levels[j], d.half_width[j], d.basis[j] = L, hw, int(basis[j])
        d.half_l[j], d.offset[j], d.shift[j] = float(hl), float(of), float(sh)
        lo0, hi0 = _f2o(np.float32(-100.0)), _f2o(np.float32(100.0))
        assert level(_o2f(lo0), hl, of, sh, hw) == 0 and level(_o2f(hi0), hl, of, sh, hw) == L - 1
        for k in range(1, L):
            lo, hi = lo0, hi0                                   # level(lo) < k <= level(hi)
            while hi - lo > 1:
                mid = (lo + hi) // 2
                if level(_o2f(mid), hl, of, sh, hw) >= k:
                    hi = mid
                else:
                    lo = mid
            if tanh_impl == "torch":
                # a threshold describes the library only where its tanh is monotone: check every float within 64 ulps
                win = np.array([_o2f(hi + i) for i in range(-64, 65)], dtype=np.float32)
                ge = _levels_torch(win, hl, of, sh, hw) >= k
                if not (np.all(~ge[:64]) and np.all(ge[64:])):
                    raise RuntimeError(f"FSQ(tanh_impl='torch'): this host's torch.tanh is not monotone around the level-{k} "
                                       f"threshold of dimension {j} ({_o2f(hi)!r}); use tanh_impl='exact'")
            d.thresh[j][k - 1] = float(_o2f(hi))
        for k in range(L - 1, 8):
            d.thresh[j][k] = float("inf")
    return d


class FSQ(nn.Module):
    """Finite scalar quantiser.  forward(z (..., d) fp32) -> (codes (..., d), indices (...) int32)."""

    def __init__(self, levels: List[int], dim=None, num_codebooks=1, tanh_impl=None, **unsupported):
        super().__init__()
        if (dim is not None and dim != len(levels)) or num_codebooks != 1:
            raise NotImplementedError("FSQ: projections / multiple codebooks are not used by the PreEncoder")
        self.levels = list(levels)
        self.codebook_dim = self.dim = len(levels)
        self.num_codebooks = 1
        self.codebook_size = int(np.prod(levels))
        lv, _, _, _, half_width, basis = fsq_constants(self.levels)
        self.register_buffer("_levels", lv, persistent=False)
        self.register_buffer("_basis", basis, persistent=False)
        idx = torch.arange(self.codebook_size)
        lvl = (idx.unsqueeze(-1) // basis) % lv
        self.register_buffer("implicit_codebook", (lvl - half_width) / half_width, persistent=False)
        import os
        self.tanh_impl = tanh_impl or os.environ.get("MQGAN_FSQ_TANH", "exact")
        self._desc = make_fsq_desc(self.levels, self.tanh_impl)

    def forward(self, z: torch.Tensor):
        if z.shape[-1] != self.dim:
            raise AssertionError(f"expected dimension of {self.dim} but found dimension of {z.shape[-1]}")
        codes, idx = ops.FsqFn.apply(z.float().contiguous(), self._desc)   # fp32 forced (quantizer.py:233-240)
        return codes, idx

    def indices_to_codes(self, indices: torch.Tensor) -> torch.Tensor:
        return ops.fsq_decode(indices, self._desc)

    def codes_to_indices(self, codes: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("use forward(); the reference only calls codes_to_indices from there")


class L2Quantizer(nn.Module):
    """Learned-codebook nearest-neighbour quantiser (the reference has no such
    module, so the arithmetic contract is the one in oracle/vq_ref.c).  forward -> (codes, indices)
    with a straight-through gradient; ``ema_update`` applies the standard EMA codebook update."""

    def __init__(self, codebook_size: int, dim: int, decay: float = 0.99, eps: float = 1e-5):
        super().__init__()
        if dim % 4:
            raise ValueError("L2Quantizer: dim must be a multiple of 4")
        self.codebook_size, self.dim, self.decay, self.eps = codebook_size, dim, decay, eps
        self.register_buffer("codebook", torch.randn(codebook_size, dim))
        self.register_buffer("cluster_size", torch.zeros(codebook_size))
        self.register_buffer("embed_sum", self.codebook.clone())

    @torch.no_grad()
    def indices(self, x: torch.Tensor) -> torch.Tensor:
        """Nearest code per vector.  d % 8 == 0: bf16 screening + exact f32 re-rank (csrc/vq_screen.hip); otherwise the
        f32-MFMA search.  Same result either way, bit for bit."""
        flat = x.reshape(-1, self.dim).contiguous()
        if ops.l2_screen_supported(flat.shape[0], self.codebook_size, self.dim):
            ver = (self.codebook.data_ptr(), self.codebook._version)
            if getattr(self, "_prepared", None) is None or self._prepared[0] != ver:
                self._prepared = (ver, ops.l2_screen_prepare(self.codebook))
            return ops.l2_argmin_screened(flat, self.codebook, self._prepared[1]).view(x.shape[:-1])
        return ops.l2_argmin(flat, self.codebook).view(x.shape[:-1])

    def forward(self, x: torch.Tensor):
        idx = self.indices(x)
        codes = torch.nn.functional.embedding(idx.long(), self.codebook)      # gather (data movement only)
        return x + (codes - x).detach(), idx

    @torch.no_grad()
    def ema_update(self, x: torch.Tensor, idx: torch.Tensor):
        ops.codebook_ema_update(x.reshape(-1, self.dim).contiguous(), idx.reshape(-1).to(torch.int32).contiguous(),
                                self.cluster_size, self.embed_sum, self.codebook, self.decay, self.eps)
        self._prepared = None            # the kernel rewrote the codebook behind torch's version counter
