"""ctypes front-end of oracle/vq_ref.c + the host-side FSQ constants.

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
import ctypes
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "libvq_ref.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvq_ref.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def fsq_consts(levels):
    """Per-dimension constants, produced with the same torch ops/dtypes as
    quantizer.py:68-72,111-113,132 so that they are bit-identical to the reference's."""
    import torch
    lv = torch.tensor(levels, dtype=torch.int32)
    half_l = (lv - 1) * (1 + 1e-3) / 2
    offset = torch.where(lv % 2 == 0, 0.5, 0.0)
    shift = (offset / half_l).atanh()
    half_width = lv // 2
    basis = torch.cumprod(torch.tensor([1] + list(levels[:-1])), dim=0, dtype=torch.int32)
    return dict(levels=lv.numpy().astype(np.int32), half_l=half_l.numpy().astype(np.float32),
                offset=offset.numpy().astype(np.float32), shift=shift.numpy().astype(np.float32),
                half_width=half_width.numpy().astype(np.int32), basis=basis.numpy().astype(np.int32))


def fsq_forward(z: np.ndarray, levels):
    c = fsq_consts(levels)
    z = np.ascontiguousarray(z, np.float32)
    d = z.shape[-1]
    n = z.size // d
    codes = np.empty_like(z)
    idx = np.empty(z.shape[:-1], np.int32)
    lib().fsq_ref_forward(_p(z, ctypes.c_float), ctypes.c_int64(n), ctypes.c_int(d), _p(c["half_l"], ctypes.c_float),
                          _p(c["offset"], ctypes.c_float), _p(c["shift"], ctypes.c_float),
                          _p(c["half_width"], ctypes.c_int32), _p(c["basis"], ctypes.c_int32),
                          _p(codes, ctypes.c_float), _p(idx, ctypes.c_int32))
    return codes, idx


def fsq_decode(idx: np.ndarray, levels):
    c = fsq_consts(levels)
    idx = np.ascontiguousarray(idx, np.int64)
    d = len(levels)
    codes = np.empty(idx.shape + (d,), np.float32)
    lib().fsq_ref_decode(_p(idx, ctypes.c_int64), ctypes.c_int64(idx.size), ctypes.c_int(d),
                         _p(c["levels"], ctypes.c_int32), _p(c["half_width"], ctypes.c_int32),
                         _p(c["basis"], ctypes.c_int32), _p(codes, ctypes.c_float))
    return codes


def l2_codebook_norms(cb: np.ndarray):
    cb = np.ascontiguousarray(cb, np.float32)
    cn = np.empty(cb.shape[0], np.float32)
    lib().l2_codebook_norms_ref(_p(cb, ctypes.c_float), ctypes.c_int(cb.shape[0]), ctypes.c_int(cb.shape[1]),
                                _p(cn, ctypes.c_float))
    return cn


def l2_argmin(x: np.ndarray, cb: np.ndarray, threads: int = 1, return_dist=False):
    """idx[n] = argmin_k |x_n - c_k|^2 under the arithmetic defined in vq_ref.c."""
    x = np.ascontiguousarray(x, np.float32)
    cb = np.ascontiguousarray(cb, np.float32)
    n, d = x.shape
    k = cb.shape[0]
    cn = l2_codebook_norms(cb)
    idx = np.empty(n, np.int32)
    dist = np.empty(n, np.float32)
    L = lib()

    def run(r):
        L.l2_argmin_ref(_p(x, ctypes.c_float), _p(cb, ctypes.c_float), _p(cn, ctypes.c_float), ctypes.c_int64(r[0]),
                        ctypes.c_int64(r[1]), ctypes.c_int(k), ctypes.c_int(d), _p(idx, ctypes.c_int32),
                        _p(dist, ctypes.c_float))

    threads = max(1, min(threads, n))
    bounds = np.linspace(0, n, threads + 1).astype(np.int64)
    ranges = [(int(bounds[i]), int(bounds[i + 1])) for i in range(threads)]
    if threads == 1:
        run(ranges[0])
    else:
        with ThreadPoolExecutor(threads) as ex:   # ctypes releases the GIL
            list(ex.map(run, ranges))
    return (idx, dist) if return_dist else idx


def codebook_ema(x: np.ndarray, idx: np.ndarray, cluster_size: np.ndarray, embed_sum: np.ndarray, decay=0.99, eps=1e-5):
    """One EMA codebook update under the arithmetic and summation order defined in vq_ref.c (section 3 of its
    header).  Returns new (cluster_size, embed_sum, codebook); inputs are not modified."""
    x = np.ascontiguousarray(x, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    n, d = x.shape
    k = cluster_size.shape[0]
    cs = np.array(cluster_size, np.float32, copy=True)
    es = np.array(embed_sum, np.float32, copy=True).reshape(k, d)
    cb = np.empty((k, d), np.float32)
    ws = np.empty((k, d), np.float32)
    lib().codebook_ema_ref(_p(x, ctypes.c_float), _p(idx, ctypes.c_int32), ctypes.c_int64(n), ctypes.c_int(k), ctypes.c_int(d),
                           ctypes.c_float(decay), ctypes.c_float(eps), _p(cs, ctypes.c_float), _p(es, ctypes.c_float),
                           _p(cb, ctypes.c_float), _p(ws, ctypes.c_float))
    return cs, es, cb
