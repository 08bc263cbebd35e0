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


def _stat(line):
    """With MQGAN_TEST_STATS=<file> every element-wise check appends what it measured (how the bounds were chosen)."""
    import os
    path = os.environ.get("MQGAN_TEST_STATS")
    if path:
        with open(path, "a") as f:
            f.write(line + "\n")


def assert_elem(a, b, tol, cmax, msg="", edges=None, taps=False):
    """Relative L2 AND element-wise bounds for the Winograd / polyphase / nested forms, whose failure modes are LOCAL (a wrong
    edge tile, a mis-handled causal column): a relative L2 norm over the whole tensor absorbs those.

      * ||a - b|| <= tol ||b||                              (as assert_rel)
      * max |a - b| <= cmax * rms(b)                        (no element is off by more than the form's own rounding tail)
      * edges = (first_h, last_h, first_w, last_w) for channels-last (N, H, W, C) tensors: the first / last rows and columns
        - the ragged last tile, the padding columns of a causal filter - each meet both bounds on their own (relative L2 with
        a factor 2: a slice has fewer elements to average over)
      * taps=True for (K, C, R, S) weight gradients: every tap (r, s) meets the relative L2 bound on its own
    """
    a = a.detach().double().cpu() if isinstance(a, torch.Tensor) else torch.as_tensor(a).double()
    b = b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    a = a.reshape(b.shape)
    rms = float(b.pow(2).mean().sqrt()) + 1e-30
    e = float((a - b).norm() / (b.norm() + 1e-30))
    m = float((a - b).abs().max()) / rms
    _stat(f"{msg}\trel_l2={e:.3e}\tmax_over_rms={m:.3e}\ttol={tol:.1e}\tcmax={cmax:.1e}")
    assert e <= tol, f"{msg}: relative L2 error {e:.3e} > {tol:.1e}"
    assert m <= cmax, f"{msg}: largest element error {m:.3e} of the rms > {cmax:.1e}"
    parts = []
    if edges is not None and b.dim() == 4:
        fh, lh, fw, lw = edges
        H, W = b.shape[1], b.shape[2]
        if fh:
            parts.append((f"first {fh} row(s)", (slice(None), slice(0, min(fh, H)))))
        if lh:
            parts.append((f"last {lh} row(s)", (slice(None), slice(max(H - lh, 0), H))))
        if fw:
            parts.append((f"first {fw} column(s)", (slice(None), slice(None), slice(0, min(fw, W)))))
        if lw:
            parts.append((f"last {lw} column(s)", (slice(None), slice(None), slice(max(W - lw, 0), W))))
    if taps and b.dim() == 4:
        for r in range(b.shape[2]):
            for s_ in range(b.shape[3]):
                parts.append((f"tap ({r},{s_})", (slice(None), slice(None), r, s_)))
    for name, sl in parts:
        pa, pb = a[sl], b[sl]
        if pb.numel() == 0:
            continue
        if float(pb.norm()) == 0.0:
            assert float((pa - pb).abs().max()) <= cmax * rms, f"{msg}, {name}: nonzero where the reference is zero"
            continue
        pe = float((pa - pb).norm() / pb.norm())
        pm = float((pa - pb).abs().max()) / rms
        _stat(f"{msg} [{name}]\trel_l2={pe:.3e}\tmax_over_rms={pm:.3e}")
        assert pe <= 2 * tol, f"{msg}, {name}: relative L2 error {pe:.3e} > {2 * tol:.1e}"
        assert pm <= cmax, f"{msg}, {name}: largest element error {pm:.3e} of the rms > {cmax:.1e}"


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


def check_grads_f64_yardstick(got: dict, ref32: dict, ref64: dict, msg="", bar=1e-4, factor=2.0):
    """North star: gradients within 1e-4.  Where a fixture comparison needs more than that (G4 / G5 were held at 3e-4 / 5e-4),
    the float64 yardstick says whether the HIP path or the f32 fixture itself is the one that is off: per tensor, with
    e(a) = max |a - ref64| / max |ref64|, the HIP gradient must satisfy  e(hip) <= 1e-4  OR  e(hip) <= 2 e(ref32)  - i.e. be no
    further from float64 than twice what the reference's own f32 CPU arithmetic is.  Returns the worst (e_hip, e_ref32)."""
    worst = (0.0, 0.0)
    for k, g64 in ref64.items():
        if k not in got or k not in ref32:
            continue
        g64 = g64.detach().double().cpu()
        scale = max(float(g64.abs().max()), 1e-30)
        e_hip = float((got[k].detach().double().cpu().reshape(g64.shape) - g64).abs().max()) / scale
        e_ref = float((torch.as_tensor(ref32[k]).double().reshape(g64.shape) - g64).abs().max()) / scale
        _stat(f"{msg} f64 yardstick {k}\te_hip={e_hip:.3e}\te_ref32={e_ref:.3e}")
        assert e_hip <= bar or e_hip <= factor * e_ref + 1e-9, \
            f"{msg} {k}: HIP is {e_hip:.2e} of the largest element from float64, the f32 reference {e_ref:.2e} (bar {bar:.0e} or {factor} x)"
        if e_hip > worst[0]:
            worst = (e_hip, e_ref)
    return worst
