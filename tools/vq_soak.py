#!/usr/bin/env python3
"""Randomised soak of the screened codebook search against the f32 kernel (both on the GPU, so it is fast): random shapes
(rows 1..40000, codes 1..9000, d a multiple of 8 up to 512: what the screened search takes), random scales, a few NaN / huge rows, repeated launches on the same
workspace.  Any difference in indices or distances is a bug (the LDS-DMA staging of l2_screen_kernel is the kind of code where a
missing wait shows up once in a thousand tiles).  usage: python tools/vq_soak.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    n = int(rng.choice([1, 7, 127, 128, 129, 1000, 4096, 16384, int(rng.integers(1, 40000))]))
    k = int(rng.choice([1, 5, 127, 128, 129, 1000, 1024, 2048, int(rng.integers(1, 9000))]))
    d = 8 * int(rng.choice([1, 3, 8, 9, 32, 64, int(rng.integers(1, 65))]))
    sx, sc = float(10.0 ** rng.uniform(-6, 6)), float(10.0 ** rng.uniform(-6, 6))
    x = torch.randn(n, d, device="cuda") * sx
    cb = torch.randn(k, d, device="cuda") * sc
    if rng.random() < 0.2 and n > 3:
        x[int(rng.integers(0, n)), int(rng.integers(0, d))] = float("nan")
        x[int(rng.integers(0, n))] = 1e30
    if rng.random() < 0.2 and k > 8:                      # clustered codes: many near-ties
        cb[k // 2:] = cb[:k - k // 2] + torch.randn(k - k // 2, d, device="cuda") * sc * 1e-4
    prep = ops.l2_screen_prepare(cb)
    ws = ops._ws_bytes(ops.L().mq_l2_screen_workspace(n, k, d), x)
    i0, d0 = ops.l2_argmin(x, cb, return_dist=True)
    for rep in range(3):
        i1, d1 = ops.l2_argmin_screened(x, cb, prep, return_dist=True, workspace=ws)
        same = torch.equal(i0, i1) and torch.equal(d0.view(torch.int32), d1.view(torch.int32))
        if not same:
            bad += 1
            w = (i0 != i1).nonzero().flatten()[:5].tolist()
            print(f"MISMATCH case {c} rep {rep}: n={n} k={k} d={d} sx={sx:.2e} sc={sc:.2e} rows {w}", flush=True)
            break
    if (c + 1) % 25 == 0:
        print(f"{c + 1} cases, {bad} mismatches", flush=True)
print("SOAK", "FAILED" if bad else "OK", f"{cases} cases x 3 launches, {bad} mismatches")
sys.exit(1 if bad else 0)
