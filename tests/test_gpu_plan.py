"""-m gpu: the planned convolution calls as a CONSUMER OF THE HEADER would use them - ctypes only, torch just owns the memory
(INTEGRATION.md section 2): mq_conv_plan picks the form, mq_conv_fprop_planned / mq_conv_backward_planned run it, and the results
equal those of the same calls under a configuration with every reduced form switched off (the direct kernels)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_elem  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from mqgan_amd import _lib
    return _lib, _lib.load()


CASES = [
    # name, desc kwargs, leaky, expected form
    ("refiner 3x3", dict(N=2, H=12, W=15, C=128, K=144, R=3, S=3, P=12, Q=15, stride_h=1, stride_w=1, pad_h=1, pad_w=1), False, "winograd2d"),
    ("patch-D 5x5 s2", dict(N=2, H=24, W=40, C=128, K=128, R=5, S=5, P=12, Q=20, stride_h=2, stride_w=2, pad_h=2, pad_w=2), True, "polyphase-winograd2d"),
    ("multi-bin 3x5 g2", dict(N=4, H=10, W=37, C=128, K=128, R=3, S=5, P=10, Q=37, stride_h=1, stride_w=1, pad_h=1, pad_w=2, groups=2), True,
     "winograd-rows"),
    ("multi-bin 3x3 s(1,2)", dict(N=4, H=10, W=36, C=128, K=160, R=3, S=3, P=10, Q=18, stride_h=1, stride_w=2, pad_h=1, pad_w=1, groups=2), True,
     "winograd-rows"),
    ("decoder causal k7", dict(N=2, H=1, W=45, C=128, K=128, R=1, S=7, P=1, Q=45, stride_h=1, stride_w=1, pad_h=0, pad_w=6), False, "winograd-columns"),
    ("patch-D 3x3 s(2,1)", dict(N=2, H=9, W=22, C=128, K=128, R=3, S=3, P=5, Q=22, stride_h=2, stride_w=1, pad_h=1, pad_w=1), True, "winograd-columns"),
    # a layer no reduced form takes, with batch groups: the direct form's bias-only call sums one group at a time
    ("narrow band layer g4", dict(N=8, H=10, W=36, C=32, K=48, R=3, S=3, P=10, Q=18, stride_h=1, stride_w=2, pad_h=1, pad_w=1, groups=4), True, "direct"),
]


@pytest.mark.parametrize("name,kw,leaky,form", CASES, ids=[c[0] for c in CASES])
def test_planned_calls_through_ctypes_equal_the_direct_form(lib, name, kw, leaky, form):
    _lib, L = lib
    torch.manual_seed(3)
    d = _lib.ConvDesc(dil_h=1, dil_w=1, groups=kw.get("groups", 1), precision=0, **{k: v for k, v in kw.items() if k != "groups"})
    G = d.groups
    x = torch.nn.functional.leaky_relu(torch.randn(d.N, d.H, d.W, d.C, device=DEV), 0.2)
    w_krsc = torch.randn(G * d.K, d.R * d.S, d.C, device=DEV) * 0.05          # (taps, channels) rows; channel blocks of 16 inside
    bias = torch.randn(G * d.K, device=DEV)
    dy = torch.randn(d.N, d.P, d.Q, d.K, device=DEV)
    mask = torch.zeros(d.N, d.Q, dtype=torch.uint8, device=DEV)
    mask[-1, d.Q // 2:] = 1
    gmask = torch.zeros(d.N, d.W, dtype=torch.uint8, device=DEV)
    gmask[-1, d.W // 2:] = 1
    # CRSK twin of the weights for the direct input gradient: undo the channel blocking, transpose, block again
    CB = 16
    plain = w_krsc.view(G, d.K, d.C // CB, d.R * d.S, CB).permute(0, 1, 3, 2, 4).reshape(G, d.K, d.R * d.S, d.C)
    w_crsk = plain.permute(0, 3, 2, 1).reshape(G, d.C, d.R * d.S, d.K // CB, CB).permute(0, 1, 3, 2, 4).contiguous().view(G * d.C, d.R * d.S, d.K)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    out = {}
    for leg in ("planned", "planned-no-keep", "direct"):       # with and without the forward -> backward keep buffer
        use_keep = leg == "planned"
        cfg = _lib.PlanCfg()
        L.mq_plan_cfg_default(C.byref(cfg))
        if leg == "direct":
            cfg.wino_minc = cfg.s2d_minc = cfg.wino1d_minc = cfg.winow_minc = 0
            cfg.wino_minc_explicit = 1
        p = _lib.ConvPlan()
        assert L.mq_conv_plan(C.byref(d), int(leaky), C.byref(cfg), C.byref(p)) == 0, L.mq_last_error()
        assert L.mq_form_name(p.form).decode() == (form if leg != "direct" else "direct")
        # 0xFF bytes = NaNs: whatever a planned call reads from its buffers it must have written itself
        nws = (max(p.ws_fprop, p.ws_backward, 256) + 255) // 256 * 256
        arena = torch.full((nws + 4096 + max(p.keep_bytes, 256),), 0xFF, dtype=torch.uint8, device=DEV)
        ws, guard = arena[:nws], arena[nws:nws + 4096]
        guard.fill_(0x5A)
        keep = arena[nws + 4096:] if (use_keep and p.keep_bytes) else None
        y = torch.empty(d.N, d.P, d.Q, d.K, device=DEV)
        rc = L.mq_conv_fprop_planned(C.byref(p), P(x), P(w_krsc), P(bias), 2 if leaky else 0, 0.2, P(mask) if leaky else None, d.P * d.Q, d.Q, d.Q,
                                     P(y), P(keep), P(ws), nws, st)
        assert rc == 0, L.mq_last_error()
        # the buffers are exactly as large as the plan says, and each leg's keep buffer sits right behind its workspace in one
        # allocation: a planned call that carved past the end of its workspace would corrupt what it handed to backward (round 4:
        # the sizing pass under-counted the bias partials of the two-dimensional forms - found by exactly this adjacency)
        ws.fill_(0xFF)
        dx, dw, db = torch.full_like(x, float("nan")), torch.full_like(w_krsc, float("nan")), torch.full_like(bias, float("nan"))
        gate = leaky                          # the leaky layers sit in gated chains: x came out of a LeakyReLU + mask layer
        rc = L.mq_conv_backward_planned(C.byref(p), P(x), P(keep), P(w_krsc), P(w_crsk), P(dy), P(x) if gate else None, 0.2,
                                        P(gmask) if gate else None, d.H * d.W, d.W, d.W, P(dx), P(dw), P(db), P(ws), nws, st)
        assert rc == 0, L.mq_last_error()
        torch.cuda.synchronize()
        assert bool((guard == 0x5A).all()), f"{leg}: a planned call wrote past the end of its workspace"
        out[leg] = (y, dx, dw, db)
        # every subset of the gradients out of the same plan.ws_backward bytes (round 4 advice: the direct form's bias-only call
        # carved groups x more scratch than the plan had sized)
        for want in ("dx", "dw", "db"):
            ws.fill_(0xFF)
            sub = {"dx": None, "dw": None, "db": None}
            sub[want] = torch.full_like({"dx": x, "dw": w_krsc, "db": bias}[want], float("nan"))
            rc = L.mq_conv_backward_planned(C.byref(p), P(x), P(keep), P(w_krsc), P(w_crsk), P(dy), P(x) if gate else None, 0.2,
                                            P(gmask) if gate else None, d.H * d.W, d.W, d.W, P(sub["dx"]), P(sub["dw"]), P(sub["db"]),
                                            P(ws), nws, st)
            assert rc == 0, f"{leg} {want} only: {L.mq_last_error()}"
            torch.cuda.synchronize()
            assert bool((guard == 0x5A).all()), f"{leg} {want} only: a planned call wrote past the end of its workspace"
            full = {"dx": dx, "dw": dw, "db": db}[want]
            if want == "dx":
                assert torch.equal(sub[want], full), f"{leg}: dx alone differs from dx with the other gradients"
            else:       # the bias sums may come from another pass (dy's column sums instead of the transform's partial sums)
                assert_elem(sub[want], full, 2e-6, 2e-5, f"plan ctypes {name} {leg} {want} alone")
    # the filters-only keep buffer (frozen weights: the generator step's passes through the discriminators): the row / column forms
    # carry their transformed CRSK filters in keep_uc_bytes bytes and nothing else; dx must be what the full keep buffer gives
    p = _lib.ConvPlan()
    cfg = _lib.PlanCfg()
    L.mq_plan_cfg_default(C.byref(cfg))
    assert L.mq_conv_plan(C.byref(d), int(leaky), C.byref(cfg), C.byref(p)) == 0, L.mq_last_error()
    assert (p.keep_uc_bytes > 0) == (form in ("winograd-rows", "winograd-columns")) and p.keep_filters_only == 0
    if p.keep_uc_bytes:
        p.keep_filters_only = 1
        nws = (max(p.ws_fprop, p.ws_backward, 256) + 255) // 256 * 256
        arena = torch.full((nws + 4096 + p.keep_uc_bytes + 4096,), 0xFF, dtype=torch.uint8, device=DEV)
        ws, g1, keep, g2 = arena[:nws], arena[nws:nws + 4096], arena[nws + 4096:nws + 4096 + p.keep_uc_bytes], arena[nws + 4096 + p.keep_uc_bytes:]
        g1.fill_(0x5A); g2.fill_(0x5A)
        y = torch.empty(d.N, d.P, d.Q, d.K, device=DEV)
        assert L.mq_conv_fprop_planned(C.byref(p), P(x), P(w_krsc), P(bias), 2 if leaky else 0, 0.2, P(mask) if leaky else None, d.P * d.Q, d.Q,
                                       d.Q, P(y), P(keep), P(ws), nws, st) == 0, L.mq_last_error()
        ws.fill_(0xFF)
        dx = torch.full_like(x, float("nan"))
        gate = leaky
        assert L.mq_conv_backward_planned(C.byref(p), None, P(keep), P(w_krsc), None, P(dy), P(x) if gate else None, 0.2,
                                          P(gmask) if gate else None, d.H * d.W, d.W, d.W, P(dx), None, None, P(ws), nws, st) == 0, L.mq_last_error()
        torch.cuda.synchronize()
        assert bool((g1 == 0x5A).all()) and bool((g2 == 0x5A).all()), "filters-only keep: wrote outside the workspace or the keep buffer"
        assert torch.equal(y, out["planned"][0]) and torch.equal(dx, out["planned"][1])
        dwn = torch.empty_like(w_krsc)          # a weight gradient cannot come out of a filters-only keep buffer without x
        assert L.mq_conv_backward_planned(C.byref(p), None, P(keep), P(w_krsc), None, P(dy), None, 1.0, None, 1, 0, 1, None, P(dwn), None,
                                          P(ws), nws, st) != 0
    for leg in ("planned", "planned-no-keep"):
        for nm, a, b in zip(("y", "dx", "dw", "db"), out[leg], out["direct"]):
            assert_elem(a, b, 2e-5, 2e-4, f"plan ctypes {name} {leg} {nm}")
    for a, b in zip(out["planned"], out["planned-no-keep"]):
        assert torch.equal(a, b)                                  # the keep buffer holds exactly what backward would recompute
    if leaky:
        assert float(out["planned"][0][-1, :, d.Q // 2:].abs().max()) == 0.0        # padded output frames
        assert float(out["planned"][1][-1, :, d.W // 2:].abs().max()) == 0.0        # padded input frames of the gated gradient


def test_fused_winograd_prototype_matches_float64(lib, variants_lib):
    """tools/microbench/winofused.hip (round-5 prototype, never chosen by mq_conv_plan; since round 6 compiled into the
    test-variant library only - the product library neither contains nor exports it): F(4x5, 3x3) with both transforms inside the GEMM kernel,
    on a ragged image (H, W not multiples of the tile) with 64 output channels - against a float64 convolution at the bounds the
    separate Winograd sequence meets (relative L2 2e-5, no element off by more than 1e-4 of the rms), first / last rows and
    columns on their own."""
    _lib, Lp = lib
    assert not hasattr(Lp, "mq_winofused_fprop") and not hasattr(Lp, "mq_winon_weight_transform")      # not in the product library
    L = variants_lib
    torch.manual_seed(5)
    N, H, W, Cc, K = 3, 22, 33, 48, 64
    x = torch.randn(N, H, W, Cc, device=DEV)
    w = torch.randn(K, Cc, 3, 3, device=DEV) / (3 * Cc ** 0.5)
    bias = torch.randn(K, device=DEV)
    from mqgan_amd import ops
    wk, _ = ops.weight_prep(w, None, False)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    U = torch.empty(42 * K * Cc, device=DEV)
    assert L.mq_winon_weight_transform(P(wk), K, Cc, 0, P(U), 4, 5, st) == 0, L.mq_last_error()
    Ut = torch.full_like(U, float("nan"))
    y = torch.full((N, H, W, K), float("nan"), device=DEV)
    assert L.mq_winofused_fprop(P(x), P(U), P(bias), N, H, W, Cc, K, P(Ut), P(y), st) == 0, L.mq_last_error()
    ref = torch.nn.functional.conv2d(x.double().permute(0, 3, 1, 2).cpu(), w.double().cpu(), bias.double().cpu(), padding=1).permute(0, 2, 3, 1)
    assert_elem(y, ref, 2e-5, 1e-4, "fused Winograd prototype", edges=(1, 2, 1, 3))
    assert L.mq_winofused_fprop(P(x), P(U), P(bias), N, H, W, Cc, 80, P(Ut), P(y), st) != 0            # K > 64: refused, not mis-run
