"""Achieved HBM bandwidth of the pointwise (activation / dropout / residual / mask) kernels at the refiner's and the
discriminators' shapes.   python tools/pw_micro.py [bf16|f32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge

ge.build()
from mqgan_amd import ops

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
esz = 2 if dt == torch.bfloat16 else 4
dev = torch.device("cuda:0")


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


CASES = [("c3 L0 (32,256,128,96)", (32, 256, 128, 96)), ("c3 L1 (32,128,128,192)", (32, 128, 128, 192)),
         ("c3 L3 (32,32,128,768)", (32, 32, 128, 768)), ("c2 L0 (32,256,80,64)", (32, 256, 80, 64))]
for name, shp in CASES:
    N, T, F, Cc = shp
    h = torch.randn(shp, device=dev).to(dt)
    r = torch.randn(shp, device=dev).to(dt)
    g = torch.randn(shp, device=dev).to(dt)
    mask = (torch.rand(N, T, device=dev) > 0.8).to(torch.uint8)
    n = h.numel()
    for label, kw, has_post in [("aptx+drop", dict(act=ops.ACT_APTX, p_drop=0.1, training=True), False),
                                ("aptx+drop+res+mask", dict(act=ops.ACT_APTX, p_drop=0.1, training=True, r_post=r, mask=mask, mask_post=True, dmap=(F, 1, 1 << 30)), True),
                                ("leaky+mask", dict(act=ops.ACT_LEAKY, mask=mask, mask_post=True, dmap=(F, 1, 1 << 30)), False)]:
        hh = h.clone().requires_grad_(True)
        if has_post:
            kw = dict(kw)
            kw["r_post"] = r.clone().requires_grad_(True)
        y = ops.pointwise(hh, **kw)
        tf = timeit(lambda: ops.pointwise(hh, **kw))
        tb = timeit(lambda: y.backward(g, retain_graph=True))
        bf = n * esz * (3 if has_post else 2)
        bb = n * esz * (4 if has_post else 3)
        print(f"{name:26s} {label:20s} fwd {tf:7.3f} ms {bf / tf / 1e9:6.2f} TB/s   bwd {tb:7.3f} ms {bb / tb / 1e9:6.2f} TB/s", flush=True)
