#!/usr/bin/env python3
"""Do an MFMA-bound GEMM kernel and an HBM-bound streaming kernel overlap when they are launched on two HIP streams?  (Round 6: the
question behind running the weight gradients of the backward pass on a side stream.)  Times, on one MI355X: a Winograd-domain
weight-gradient GEMM alone, a streaming pass alone, and both concurrently on two streams.  usage: python tools/overlap_probe.py"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import ops  # noqa: E402
from mqgan_amd.ops import L, check, fptr, ptr  # noqa: E402

dev = "cuda"
# the dense Winograd-domain product of the refiner's 512 -> 512 layer: 42 groups, 9216 rows each
G, M, Cc, K = 42, 9216, 512, 512
d = ops.conv_desc(G * M // (1 * 96), 1, 96, Cc, K, 1, 1, groups=G, precision="f32") if False else None
x = torch.randn(G * M, Cc, device=dev)
dy = torch.randn(G * M, K, device=dev)
desc = ops.conv_desc(G * 96, 1, 96, Cc, K, 1, 1, groups=G, precision="f32")          # N = 42 * 96 images of 1 x 96 pixels: 9216 rows per group
dw = torch.empty(G * K * Cc, device=dev)
nb = L().mq_conv_wgrad_workspace(C.byref(desc))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
big = torch.randn(1 << 28, device=dev)            # 1 GiB: the streaming pass reads two and writes one
out = torch.empty_like(big)


def gemm(st):
    check(L().mq_conv_wgrad(C.byref(desc), fptr(x), fptr(dy), fptr(dw), ptr(ws), nb, C.c_void_p(st.cuda_stream)), "wgrad")


def stream_pass(st):
    check(L().mq_axpby(fptr(big), fptr(big), 0.5, 0.5, big.numel(), fptr(out), C.c_void_p(st.cuda_stream)), "axpby")


# argv[1]: "equal" (default) or "prio" - the streaming pass on a HIGH-priority stream, the GEMM on a low-priority one.  The streaming
# pass's grid is capped by MQGAN_EW_CAP (read once per process by the library): with a few hundred looping blocks it can only ever hold
# a slot or two per CU, whatever the dispatcher does.
mode = sys.argv[1] if len(sys.argv) > 1 else "equal"
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
sa = torch.cuda.Stream(priority=0) if mode == "equal" else torch.cuda.Stream(priority=0)
sb = torch.cuda.Stream(priority=0) if mode == "equal" else torch.cuda.Stream(priority=-1)


def run(fa, fb, reps=30):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if fa:
            fa(sa)
        if fb:
            fb(sb)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    run(gemm, stream_pass, 5)
a, b, ab = run(gemm, None), run(None, stream_pass), run(gemm, stream_pass)
flop = 2.0 * G * M * Cc * K
print(f"[{mode}, MQGAN_EW_CAP={os.environ.get('MQGAN_EW_CAP', 'default')}] GEMM alone {a:.3f} ms ({flop / a / 1e9:.0f} TF/s); streaming pass alone {b:.3f} ms ({3 * 4 * big.numel() / b / 1e9:.2f} TB/s); "
      f"both on two streams {ab:.3f} ms (sum {a + b:.3f}, max {max(a, b):.3f}): overlap saves {100 * (a + b - ab) / min(a, b):.0f} % of the shorter one")
