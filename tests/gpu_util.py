"""Helpers for the -m gpu parity tests (product HIP path vs oracle / golden fixtures)."""
import numpy as np
import torch

DEV = "cuda"


def npz_state_dict(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix)}


def load_golden_module(module, g, prefix):
    """strict load of a golden state_dict (reference key names) into a product module, then to the GPU."""
    sd = npz_state_dict(g, prefix)
    module.load_state_dict(sd, strict=True)
    return module.to(DEV)


def t(a, requires_grad=False):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    if requires_grad:
        x.requires_grad_(True)
    return x


def close(a, b, rtol=1e-4, atol=1e-5, msg=""):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=msg)


def rel_err(a, b):
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def assert_rel(a, b, tol=1e-4, msg=""):
    e = rel_err(a, b)
    assert e <= tol, f"{msg}: relative L2 error {e:.3e} > {tol:.1e}"


def check_param_grads(module, g, prefix, tol=2e-4, skip=()):
    for name, p in module.named_parameters():
        key = prefix + name
        if key not in g.files or name in skip:
            continue
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        want = torch.from_numpy(g[key])
        scale = max(want.abs().max().item(), 1e-6)
        err = (got.detach().cpu() - want).abs().max().item()
        assert err <= tol * scale + 1e-7, f"grad {name}: max abs err {err:.3e} vs scale {scale:.3e}"
