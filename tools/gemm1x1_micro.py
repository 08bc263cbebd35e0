#!/usr/bin/env python3
"""The batch-grouped 1x1 convolution the Winograd forms run on (42 groups, one-row images): forward GEMM alone at the channel
counts of the config-2 refiner, HIP-event time -> TF/s and the HBM rate the algorithmic bytes (V read, M written) imply.
`gemm1x1_micro.py [rows_per_group [C K]]`"""
import os, sys
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402
ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.ops import L, fptr, stream, check  # noqa: E402

DEV = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 36864
G = 42
SHAPES = ((128, 128), (256, 256), (512, 512), (768, 256), (256, 512), (64, 64))
if len(sys.argv) > 3:
    SHAPES = ((int(sys.argv[2]), int(sys.argv[3])),)
for Cc, K in SHAPES:
    tw = 18
    nimg = rows // tw
    d = ops.conv_desc(G * nimg, 1, tw, Cc, K, 1, 1, groups=G, precision="f32")
    V = torch.randn(G * nimg, 1, tw, Cc, device=DEV)
    U = torch.randn(G * K, 1, Cc, device=DEV) * 0.05
    M = torch.empty(G * nimg, 1, tw, K, device=DEV)
    fn = lambda: check(L().mq_conv_fprop(C.byref(d), fptr(V), fptr(U), None, fptr(M), stream()), "mq_conv_fprop")
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    fl = 2.0 * G * nimg * tw * Cc * K
    by = 4.0 * G * nimg * tw * (Cc + K)
    print(f"C {Cc:4d} -> K {K:4d}: {ms:7.3f} ms  {fl / ms / 1e9:6.1f} TF/s  {by / ms / 1e9:6.2f} TB/s algorithmic  kernel {L().mq_last_kernel().decode()}", flush=True)
