"""One small invocation of the hot path on cuda:0, checked against the oracle / golden fixtures
(called by __graft_entry__.smoke())."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run():
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs an MI355X")
    from mqgan_amd.preencoder import PreEncoder
    from mqgan_amd.quantizer import FSQ
    from oracle import nn as O
    dev = "cuda:0"
    g = np.load(os.path.join(ROOT, "tests", "golden", "g9_encdec.npz"))
    P = O.params_from_npz(g, "sd/", requires_grad=False)
    m = PreEncoder(mel_channels=16, channels=[16, 16, 24], kernel_sizes=[3, 5], fsq_levels=[8, 5, 5, 5], dropout=0.1,
                   refiner_base_channels=8, refiner_depth=3)
    m.load_state_dict({k: v.clone() for k, v in P.items()}, strict=True)
    m = m.to(dev).eval()
    x, lens = torch.from_numpy(g["x"]), torch.from_numpy(g["lens"])
    # forward + backward on the HIP path
    xd = x.to(dev)
    rec, post = m(xd, lens.to(dev))
    (post.sum() + rec.sum()).backward()
    # oracle on the CPU
    with torch.no_grad():
        rec_o, post_o = O.preencoder_forward(P, x, lens, [8, 5, 5, 5])
        mask3 = (torch.arange(x.shape[1])[None] >= lens[:, None]).unsqueeze(1)
        idx_o = O.preencoder_encode(P, x, mask3, [8, 5, 5, 5])
    idx = m.encode(xd, mask3.to(dev))
    assert torch.equal(idx.cpu(), idx_o), "FSQ indices differ from the oracle"
    err = ((post.detach().cpu() - post_o).norm() / post_o.norm()).item()
    assert err < 1e-4, f"x_post relative error {err:.3e}"
    err = ((rec.detach().cpu() - rec_o).norm() / rec_o.norm()).item()
    assert err < 1e-4, f"x_recon relative error {err:.3e}"
    gn = sum(float(p.grad.abs().sum()) for p in m.parameters() if p.grad is not None)
    assert np.isfinite(gn) and gn > 0
    q = FSQ([8, 5, 5, 5]).to(dev)
    z = torch.randn(4096, 4) * 2
    _, i_gpu = q(z.to(dev))
    _, i_cpu = O.fsq_forward(z, [8, 5, 5, 5])
    assert torch.equal(i_gpu.cpu(), i_cpu), "FSQ kernel differs from the oracle"
    print("smoke ok: indices bit-exact, x_recon/x_post within 1e-4 of the oracle, gradients finite")
