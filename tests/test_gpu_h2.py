"""-m gpu: the bf16-storage convolution kernels (csrc/igemm_h2.hip) through the C ABI against float64 CPU convolutions.

Operands are drawn bf16-representable, so every product is exact and the f32 accumulation carries the only error:
with f32 output the kernels must meet the same 2e-5 bound as the f32 engine; with bf16 output the result must be the
correctly rounded reference to within one bf16 ulp (2^-8 relative)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import DEV, assert_elem, assert_rel  # noqa: E402
from test_gpu_kernels import _torch_conv  # noqa: E402

BF16 = torch.bfloat16


@pytest.fixture(scope="module")
def env():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()
    from mqgan_amd import ops
    return ops


H2_CASES = [
    dict(N=2, H=24, W=18, C=64, K=64, R=3, S=3, pad=(1, 1)),                                  # CB = 64, TN = 1
    dict(N=2, H=20, W=33, C=96, K=96, R=3, S=3, pad=(1, 1)),                                  # CB = 32, odd position count
    dict(N=1, H=1, W=300, C=32, K=128, R=1, S=1),                                             # linear
    dict(N=1, H=40, W=18, C=288, K=96, R=3, S=3, pad=(1, 1)),                                 # refiner ups.2.conv1 (C3 widths)
    dict(N=1, H=8, W=32, C=384, K=768, R=3, S=3, pad=(1, 1)),                                 # TN = 4, three N tiles
    dict(N=1, H=16, W=40, C=192, K=192, R=3, S=3, pad=(1, 1)),                                # TN = 3
    dict(N=2, H=16, W=20, C=64, K=128, R=5, S=5, stride=(2, 2), pad=(2, 2)),                  # stride (2,2): parity dgrad
    dict(N=2, H=9, W=21, C=32, K=64, R=3, S=3, stride=(2, 1), pad=(1, 1)),                    # stride (2,1), odd H
    dict(N=1, H=12, W=40, C=96, K=32, R=7, S=7, stride=(2, 2), pad=(3, 3)),                   # 7x7 s2, CB = 32
    dict(N=16, H=2, W=40, C=32, K=64, R=3, S=5, stride=(1, 2), pad=(1, 2), groups=8),         # multibin: 8 groups
    dict(N=1, H=5, W=37, C=64, K=32, R=3, S=3, pad=(2, 2), dil=(2, 2)),                       # dilation
    dict(N=1, H=1, W=70, C=128, K=160, R=1, S=7, pad=(0, 6), out_hw=(1, 70)),                 # causal 1-D, ragged N tile
    dict(N=3, H=30, W=90, C=64, K=64, R=3, S=3, pad=(1, 1)),                                  # Q >= 64: one-wrap pixel walk
    # the shared-row weight gradient (conv_wgrad_h2s_kernel: C % 128 == 0, K % 192 == 0, three stride-1 column taps); the 384 -> 768
    # case above takes it too
    dict(N=2, H=6, W=144, C=128, K=192, R=3, S=3, pad=(1, 1)),                                # one-wrap walk over the padded width
    dict(N=4, H=1, W=200, C=128, K=192, R=1, S=3, pad=(0, 1), groups=2),                      # one tap row, two groups
    dict(N=1, H=30, W=37, C=256, K=384, R=3, S=3, pad=(1, 1)),                                # two row tiles, two chunks, 39 columns
    # the 96-channel shared-row kernel (conv_h2s96_kernel: reduction channels % 96 == 0 and % 64 != 0, three stride-1 column taps);
    # the 96 -> 96 and 288 -> 96 cases above take it too (forward, and the input gradient with 3 column tiles)
    dict(N=2, H=5, W=144, C=96, K=192, R=3, S=3, pad=(1, 1)),                                 # two column tiles; rows wrap inside a tile
    dict(N=4, H=3, W=50, C=96, K=96, R=1, S=3, pad=(0, 1), groups=2),                         # one tap row, two groups, ragged last tile
    dict(N=1, H=9, W=31, C=288, K=288, R=3, S=3, pad=(1, 1)),                                 # three chunks x three column tiles
    # strided input gradients by stride-parity class through the shared-row kernel (64-channel chunks of dy, >= 128 columns)
    dict(N=2, H=12, W=20, C=128, K=64, R=7, S=7, stride=(2, 2), pad=(3, 3)),                  # classes of 4 / 3 column taps
    dict(N=2, H=9, W=21, C=128, K=128, R=3, S=3, stride=(2, 2), pad=(1, 1)),                  # 2 taps shared, 1 tap on the per-tap kernel
]


def _ids(c):
    return "N{N}H{H}W{W}C{C}K{K}R{R}S{S}".format(**c) + f"g{c.get('groups', 1)}"


def _setup(ops, c):
    torch.manual_seed(0)
    G = c.get("groups", 1)
    rb = lambda v: v.bfloat16().double()
    x = rb(torch.randn(c["N"], c["H"], c["W"], c["C"], dtype=torch.float64))
    w = rb(torch.randn(G * c["K"], c["C"], c["R"], c["S"], dtype=torch.float64) * 0.2)
    b = torch.randn(G * c["K"], dtype=torch.float64).float().double()
    desc = ops.conv_desc(c["N"], c["H"], c["W"], c["C"], c["K"], c["R"], c["S"], stride=c.get("stride", (1, 1)),
                         pad=c.get("pad", (0, 0)), dil=c.get("dil", (1, 1)), groups=G, out_hw=c.get("out_hw"), precision="bf16")
    wd = w.float().to(DEV)
    ks, cs = [], []
    for gi in range(G):
        a, bb = ops.weight_prep(wd[gi * c["K"]:(gi + 1) * c["K"]].contiguous(), None, True)
        ks.append(a); cs.append(bb)
    wk, wc = torch.cat(ks, 0), torch.cat(cs, 0)
    return x, w, b, desc, wk, wc


@pytest.mark.parametrize("c", H2_CASES, ids=_ids)
def test_h2_fprop_dgrad_wgrad_vs_float64(env, c):
    ops = env
    x, w, b, desc, wk, wc = _setup(ops, c)
    assert ops.h2_supported(desc)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y_ref = _torch_conv(x, w, b, c)
    r = torch.randn_like(y_ref).bfloat16().double()
    (y_ref * r).sum().backward()

    xh = x.detach().to(DEV).to(BF16)
    w_h2 = ops.h2_weight_image(desc, wk, False)
    wt_h2 = ops.h2_weight_image(desc, wc, True)
    bd = b.float().to(DEV)
    y32 = ops.conv_fprop_h2(desc, xh, w_h2, bd, out_dtype=torch.float32)
    assert tuple(y32.shape) == tuple(y_ref.shape)
    assert_rel(y32, y_ref, 2e-5, "fprop (f32 out)")
    y16 = ops.conv_fprop_h2(desc, xh, w_h2, bd)
    assert y16.dtype == BF16
    on96 = (c["S"] == 3 and c.get("stride", (1, 1)) == (1, 1) and c.get("dil", (1, 1)) [1] == 1 and c["C"] % 96 == 0 and c["C"] % 64 != 0 and
            c["K"] % 96 == 0)
    assert ops.L().mq_last_kernel().decode().startswith("conv_h2s96") == on96       # exactly these take the 96-channel kernel
    want16 = y_ref.detach().float().to(BF16).float()
    got16 = y16.float().cpu()
    # one bf16 ulp is <= 2^-7 of the value; the f32 accumulation error (1e-5 of the tensor's scale) can move a value
    # across a rounding boundary
    ulp = want16.abs() * 2.0 ** -7 + 1e-5 * float(y_ref.detach().pow(2).mean().sqrt())
    assert bool(((got16 - want16).abs() <= ulp).all()), "fprop (bf16 out) is off by more than one ulp"
    rh = r.to(DEV).to(BF16)
    dx = ops.conv_dgrad_h2(desc, rh, wt_h2, out_dtype=torch.float32)
    assert_rel(dx, x.grad, 2e-5, "dgrad")
    dx16 = ops.conv_dgrad_h2(desc, rh, wt_h2)
    assert_rel(dx16.float(), x.grad, 4e-3, "dgrad (bf16 out)")
    if ops.h2_wgrad_supported(desc):
        dw, db = ops.conv_wgrad_h2(desc, xh, rh, tuple(wk.shape), True)
        shared = (c["S"] == 3 and c.get("stride", (1, 1)) == (1, 1) and c.get("dil", (1, 1))[1] == 1 and c["C"] % 128 == 0 and
                  c["K"] % 192 == 0 and desc.Q >= 14)
        # r6: what the 128-channel form does not take of the 96-channel-chunk layers goes to conv_wgrad_h2s96_kernel<RG, .>
        shared96 = (c["S"] == 3 and c.get("stride", (1, 1)) == (1, 1) and c.get("dil", (1, 1))[1] == 1 and c["C"] % 96 == 0 and
                    c["K"] % 96 == 0 and desc.Q >= 14 and not shared)
        last = ops.L().mq_last_kernel().decode()
        assert last.startswith("conv_wgrad_h2s_kernel") == shared, (last, shared)       # the shared-row kernel takes exactly these
        assert last.startswith("conv_wgrad_h2s96_kernel") == shared96, (last, shared96)
        if shared96:
            assert last.startswith("conv_wgrad_h2s96_kernel<%d, " % (2 if c["K"] % 192 == 0 else 1)), last
        # bring dw (KRSC, channel-blocked) back to the reference layout through the weight-prep backward
        G = c.get("groups", 1)
        dws = []
        for gi in range(G):
            v = torch.zeros(c["K"], c["C"], c["R"] * c["S"], device=DEV)
            dv = torch.empty_like(v)
            from mqgan_amd.ops import check, fptr, stream
            check(ops.L().mq_weight_prep_bwd(0, fptr(dw[gi * c["K"]:(gi + 1) * c["K"]].contiguous()), fptr(v), None, None, c["K"],
                                             c["C"], c["R"] * c["S"], fptr(dv), None, stream()), "mq_weight_prep_bwd")
            dws.append(dv.view(c["K"], c["C"], c["R"], c["S"]))
        assert_rel(torch.cat(dws, 0), w.grad, 2e-5, "wgrad")
        # the shared-row kernel's failure modes are LOCAL (the two zero positions per padded row, the per-tap scatter of its
        # epilogue): every tap on its own, and no element off by more than the accumulation's rounding tail
        assert_elem(torch.cat(dws, 0), w.grad, 2e-5, 2e-4, "wgrad " + ("(shared rows)" if shared else "(per tap)"), taps=True)
        assert_rel(db, b.grad, 2e-5, "bias grad")
    else:
        assert desc.Q < 16


@pytest.mark.parametrize("c", [dict(N=4, H=6, W=40, C=64, K=96, R=3, S=3, stride=(1, 2), pad=(1, 1)),
                               dict(N=4, H=6, W=40, C=96, K=96, R=3, S=3, pad=(1, 1))], ids=_ids)      # the second: conv_h2s96_kernel
def test_h2_leaky_mask_epilogue(env, c):
    ops = env
    x, w, b, desc, wk, wc = _setup(ops, c)
    y_ref = _torch_conv(x, w, b, c)                                             # (N, P, Q, K)
    mask = torch.zeros(c["N"], desc.Q, dtype=torch.uint8)
    mask[1, 13:] = 1
    mask[3, 5:] = 1
    want = F.leaky_relu(y_ref, 0.2) * (1 - mask[:, None, :, None].double())
    y = ops.conv_fprop_h2(desc, x.to(DEV).to(BF16), ops.h2_weight_image(desc, wk, False), b.float().to(DEV), mask=mask.to(DEV),
                          dmap=(desc.P * desc.Q, desc.Q, desc.Q), act=2, slope=0.2, out_dtype=torch.float32)
    assert_rel(y, want, 2e-5, "leaky + mask")
    assert float(y[1, :, 13:].abs().max()) == 0.0


GATE_CASES = [
    dict(N=4, H=6, W=40, C=64, K=96, R=3, S=3, pad=(1, 1)),                                   # stride 1
    dict(N=3, H=7, W=40, C=96, K=96, R=3, S=3, pad=(1, 1)),                                   # the 96-channel shared-row kernel
    dict(N=2, H=16, W=20, C=64, K=128, R=5, S=5, stride=(2, 2), pad=(2, 2)),                  # parity classes
    dict(N=16, H=2, W=40, C=32, K=64, R=3, S=5, stride=(1, 2), pad=(1, 2), groups=8),         # 8 groups, stride (1,2)
]


@pytest.mark.parametrize("c", GATE_CASES, ids=_ids)
@pytest.mark.parametrize("out_f32", [False, True])
def test_h2_gated_dgrad_is_dgrad_then_leaky_backward(env, c, out_f32):
    """mq_conv_dgrad_h2_gated: the producer layer's LeakyReLU / mask backward in the dgrad store, bit-for-bit the two
    passes it replaces (dgrad, then the pointwise LeakyReLU backward on the conv's input)."""
    ops = env
    x, w, b, desc, wk, wc = _setup(ops, c)
    G = c.get("groups", 1)
    torch.manual_seed(5)
    xin = F.leaky_relu(torch.randn(c["N"], c["H"], c["W"], G * c["C"] // G, device=DEV), 0.2)
    mask = torch.zeros(c["N"], c["W"], dtype=torch.uint8, device=DEV)
    mask[1, c["W"] // 3:] = 1
    mask[-1, 2:] = 1
    xin = (xin * (1 - mask[:, None, :, None].float())).to(BF16)                 # what a leaky + masked_fill layer hands on
    dy = torch.randn(c["N"], desc.P, desc.Q, c["K"], device=DEV).to(BF16)
    wt = ops.h2_weight_image(desc, wc, True)
    dmap = (c["H"] * c["W"], c["W"], c["W"])
    od = torch.float32 if out_f32 else BF16
    plain = ops.conv_dgrad_h2(desc, dy, wt, out_dtype=od)
    want = ops.pointwise_bwd_raw(plain if not out_f32 else plain, xin if not out_f32 else xin.float(), act=ops.ACT_LEAKY, mask=mask,
                                 mask_post=True, dmap=dmap)
    got = ops.conv_dgrad_h2(desc, dy, wt, out_dtype=od, gate=(xin, mask, dmap, 0.2))
    assert torch.equal(got, want)
    assert float(got[1, :, c["W"] // 3:].abs().max()) == 0.0
    nomask = ops.conv_dgrad_h2(desc, dy, wt, out_dtype=od, gate=(xin, None, dmap, 0.2))
    want2 = ops.pointwise_bwd_raw(plain, xin if not out_f32 else xin.float(), act=ops.ACT_LEAKY)
    assert torch.equal(nomask, want2)


SMALL_K_CASES = [
    dict(N=8, H=3, W=100, C=32, K=128, R=1, S=1, groups=4),                                   # the multi-bin first layer: 32 -> 128, groups
    dict(N=2, H=7, W=33, C=64, K=384, R=1, S=1),                                              # 64 -> 384: TN = 3, two column tiles
    dict(N=1, H=9, W=50, C=96, K=32, R=1, S=1),                                               # 96 -> 32: two K-steps of 32-channel chunks, TN = 1
    dict(N=2, H=5, W=41, C=256, K=160, R=1, S=1),                                             # four K-steps, ragged column tile
    dict(N=3, H=1, W=70, C=64, K=64, R=1, S=3, pad=(0, 1)),                                   # 1x3 on one column tile (TN = 1: not the shared-row kernel)
    dict(N=2, H=6, W=20, C=128, K=256, R=1, S=1, stride=(1, 2)),                              # strided forward, TN = 4
]


@pytest.mark.parametrize("c", SMALL_K_CASES, ids=_ids)
def test_h2_small_kernel_is_bit_identical(env, monkeypatch, c):
    """conv_h2_small_kernel (layers of one to four K-steps: 128-row tiles, one LDS stage, three blocks per CU) against
    conv_h2_kernel (MQGAN_H2_NO_SMALL=1): same K order, same MFMAs, same epilogue - forward (bias; LeakyReLU + mask; f32 and bf16
    store) and, where the layer is not strided, input gradient (plain and gated) bit for bit; the forward also against float64."""
    ops = env
    x, w, b, desc, wk, wc = _setup(ops, c)
    y_ref = _torch_conv(x, w, b, c)
    xh = x.to(DEV).to(BF16)
    bd = b.float().to(DEV)
    wh, wth = ops.h2_weight_image(desc, wk, False), ops.h2_weight_image(desc, wc, True)
    mask = (torch.rand(c["N"], desc.Q, device=DEV) < 0.3).to(torch.uint8)
    dmap = (desc.P * desc.Q, desc.Q, desc.Q)
    torch.manual_seed(9)
    dy = torch.randn(c["N"], desc.P, desc.Q, c["K"], device=DEV).to(BF16)
    strided = c.get("stride", (1, 1)) != (1, 1)
    gmask = (torch.rand(c["N"], c["W"], device=DEV) < 0.3).to(torch.uint8)
    gmap = (c["H"] * c["W"], c["W"], c["W"])

    def run():
        outs = [ops.conv_fprop_h2(desc, xh, wh, bd, out_dtype=torch.float32)]
        kf = ops.L().mq_last_kernel().decode()
        outs.append(ops.conv_fprop_h2(desc, xh, wh, bd))
        outs.append(ops.conv_fprop_h2(desc, xh, wh, bd, mask=mask, dmap=dmap, act=2, slope=0.2))
        kd = ""
        if not strided:
            outs.append(ops.conv_dgrad_h2(desc, dy, wth))
            kd = ops.L().mq_last_kernel().decode()
            outs.append(ops.conv_dgrad_h2(desc, dy, wth, out_dtype=torch.float32, gate=(xh, gmask, gmap, 0.2)))
        torch.cuda.synchronize()
        return outs, kf, kd
    small, kf_s, kd_s = run()
    monkeypatch.setenv("MQGAN_H2_NO_SMALL", "1")
    plain, kf_p, kd_p = run()
    monkeypatch.delenv("MQGAN_H2_NO_SMALL")
    assert kf_s.startswith("conv_h2_small_kernel<") and kf_p.startswith("conv_h2_kernel<"), (kf_s, kf_p)
    for i, (a, bb) in enumerate(zip(small, plain)):
        assert torch.equal(a, bb), f"output {i}: max {float((a.float() - bb.float()).abs().max()):.3e} ({kf_s} / {kd_s} against {kf_p} / {kd_p})"
    assert_rel(small[0], y_ref, 2e-5, "fprop (f32 out)")


@pytest.mark.parametrize("c", [dict(N=2, H=20, W=33, C=96, K=96, R=3, S=3, pad=(1, 1)), dict(N=1, H=9, W=31, C=288, K=288, R=3, S=3, pad=(1, 1)),
                               dict(N=4, H=3, W=50, C=96, K=96, R=1, S=3, pad=(0, 1), groups=2), dict(N=2, H=5, W=144, C=96, K=192, R=3, S=3, pad=(1, 1))],
                         ids=_ids)
def test_h2_96_channel_kernels_agree_bit_for_bit(env, monkeypatch, variants_lib, c):
    """conv_h2s96q_kernel (128-row tiles, no ring, three blocks per CU: the default, run through the PRODUCT library) against
    conv_h2s96_kernel (256-row tiles, double image buffer, three weight stages: the measured-and-rejected ring form, compiled into
    the test-variant library only and selected there by MQGAN_H2_96_BIG=1): same reduction order, same MFMAs, same epilogue - forward (f32, bf16,
    LeakyReLU + mask) and input gradient (plain, gated) bit for bit."""
    ops = env
    x, w, b, desc, wk, wc = _setup(ops, c)
    xh, bd = x.to(DEV).to(BF16), b.float().to(DEV)
    wh, wth = ops.h2_weight_image(desc, wk, False), ops.h2_weight_image(desc, wc, True)
    mask = (torch.rand(c["N"], desc.Q, device=DEV) < 0.3).to(torch.uint8)
    dmap = (desc.P * desc.Q, desc.Q, desc.Q)
    torch.manual_seed(12)
    dy = torch.randn(c["N"], desc.P, desc.Q, c["K"], device=DEV).to(BF16)

    def run():
        outs = [ops.conv_fprop_h2(desc, xh, wh, bd, out_dtype=torch.float32)]
        name = ops.L().mq_last_kernel().decode()
        outs.append(ops.conv_fprop_h2(desc, xh, wh, bd))
        outs.append(ops.conv_fprop_h2(desc, xh, wh, bd, mask=mask, dmap=dmap, act=2, slope=0.2))
        if c["K"] % 96 == 0 and c["K"] % 64 != 0:                  # the input gradient reduces over K
            outs.append(ops.conv_dgrad_h2(desc, dy, wth))
            outs.append(ops.conv_dgrad_h2(desc, dy, wth, out_dtype=torch.float32, gate=(xh, mask, dmap, 0.2)))
        torch.cuda.synchronize()
        return outs, name
    from mqgan_amd import _lib
    small, k_s = run()
    monkeypatch.setenv("MQGAN_H2_96_BIG", "1")
    ign, k_i = run()                                         # the product library ignores the switch: the ring form is not in it
    assert k_i == k_s
    with _lib.use_variants():
        big, k_b = run()
    monkeypatch.delenv("MQGAN_H2_96_BIG")
    assert k_s.startswith("conv_h2s96q_kernel<") and k_b.startswith("conv_h2s96_kernel<"), (k_s, k_b)
    for i, (a, bb) in enumerate(zip(small, big)):
        assert torch.equal(a, bb), f"output {i}: max {float((a.float() - bb.float()).abs().max()):.3e}"


STRIDED_DGRAD_CASES = [
    dict(N=2, H=16, W=20, C=128, K=64, R=7, S=7, stride=(2, 2), pad=(3, 3)),                  # 4 / 3 taps per class, TN = 2
    dict(N=2, H=9, W=21, C=128, K=128, R=5, S=5, stride=(2, 2), pad=(2, 2)),                  # 3 / 2 taps, odd sizes
    dict(N=16, H=2, W=40, C=192, K=64, R=3, S=5, stride=(1, 2), pad=(1, 2), groups=8),        # multibin: 8 groups, stride (1,2), TN = 3
    dict(N=2, H=10, W=22, C=256, K=64, R=3, S=3, stride=(2, 1), pad=(1, 1)),                  # stride (2,1): classes by row, 3 taps, TN = 4
    dict(N=2, H=8, W=16, C=128, K=64, R=3, S=3, stride=(2, 2), pad=(1, 1)),                   # 2 taps shared, the 1-tap classes per tap
    dict(N=2, H=6, W=300, C=128, K=64, R=5, S=5, stride=(2, 2), pad=(2, 2)),                  # wide rows: several tiles per class row
]


@pytest.mark.parametrize("c", STRIDED_DGRAD_CASES, ids=_ids)
@pytest.mark.parametrize("out_f32", [False, True])
def test_h2_strided_dgrad_classes_on_shared_rows_are_bit_identical(env, monkeypatch, c, out_f32):
    """A strided input gradient is, per stride-parity class, a stride-1 walk over dy: launch_h2 sends every class with 2 - 4 column
    taps through conv_h2s_kernel (class geometry, scattered store) and the rest through the per-tap kernel, class by class.
    Same reduction order, same MFMAs as the one-launch per-tap kernel (MQGAN_H2_NO_SHARE_PARITY=1): plain and gated results agree
    bit for bit; test_h2_fprop_dgrad_wgrad_vs_float64 holds two such shapes to float64."""
    ops = env
    x, w, b, desc, wk, wc = _setup(ops, c)
    torch.manual_seed(7)
    xin = F.leaky_relu(torch.randn(c["N"], c["H"], c["W"], c["C"], device=DEV), 0.2)
    mask = torch.zeros(c["N"], c["W"], dtype=torch.uint8, device=DEV)
    mask[1, c["W"] // 3:] = 1
    mask[-1, 2:] = 1
    xin = (xin * (1 - mask[:, None, :, None].float())).to(BF16)
    dy = torch.randn(c["N"], desc.P, desc.Q, c["K"], device=DEV).to(BF16)
    wt = ops.h2_weight_image(desc, wc, True)
    dmap = (c["H"] * c["W"], c["W"], c["W"])
    od = torch.float32 if out_f32 else BF16

    def run():
        plain = ops.conv_dgrad_h2(desc, dy, wt, out_dtype=od)
        kernel = ops.L().mq_last_kernel().decode()
        gated = ops.conv_dgrad_h2(desc, dy, wt, out_dtype=od, gate=(xin, mask, dmap, 0.2))
        torch.cuda.synchronize()
        return plain, gated, kernel
    p_s, g_s, k_s = run()
    monkeypatch.setenv("MQGAN_H2_NO_SHARE_PARITY", "1")
    p_p, g_p, k_p = run()
    monkeypatch.delenv("MQGAN_H2_NO_SHARE_PARITY")
    assert k_p.startswith("conv_h2_kernel<") and ", true, true>" in k_p, k_p
    # class launches carry a symbol of their own (3 taps at stride 2: the LAST class has one tap and stays on the per-tap kernel)
    assert (k_s.startswith("conv_h2s_kernel<") and k_s.endswith(", true>")) or c["S"] == 3 and c["stride"][1] == 2, k_s
    assert torch.equal(p_s, p_p), f"plain: max {float((p_s.float() - p_p.float()).abs().max()):.3e}"
    assert torch.equal(g_s, g_p), f"gated: max {float((g_s.float() - g_p.float()).abs().max()):.3e}"
    assert float(p_s.float().abs().sum()) > 0


def test_h2_full_size_refiner_layer_and_addressing(env):
    """C3's deepest refiner level at full size (16 x 128 x 144 pixels, 768 -> 768 channels): the H2 kernels against the
    f32-tensor bf16 engine of igemm.hip on the same bf16-representable operands (both have exact products and f32
    accumulation: agreement to f32 rounding), and an image copied to the end of the batch reproduces the first one."""
    ops = env
    torch.manual_seed(4)
    N, H, W, Cc, K = 16, 128, 144, 768, 768
    x = torch.randn(N, H, W, Cc, device=DEV).to(BF16)
    x[N - 1] = x[0]
    w = (torch.randn(K, Cc, 3, 3, device=DEV) * 0.02).to(BF16).float()
    wk, wc = ops.weight_prep(w, None, True)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, pad=(1, 1), precision="bf16")
    y = ops.conv_fprop_h2(d, x, ops.h2_weight_image(d, wk, False), None, out_dtype=torch.float32)
    assert torch.equal(y[N - 1], y[0])
    y_old = ops.conv(x.float(), wk, wc, None, d)
    scale = float(y_old.pow(2).mean().sqrt())
    assert float((y - y_old).abs().max()) / scale < 2e-5
    r = torch.randn(N, H, W, K, device=DEV).to(BF16)
    dx = ops.conv_dgrad_h2(d, r, ops.h2_weight_image(d, wc, True), out_dtype=torch.float32)
    dw, _ = ops.conv_wgrad_h2(d, x, r, tuple(wk.shape), False)
    xf = x.float().requires_grad_(True)
    wkf = wk.clone().requires_grad_(True)
    (ops.conv(xf, wkf, wc, None, d) * r.float()).sum().backward()
    assert float((dx - xf.grad).abs().max()) / float(xf.grad.pow(2).mean().sqrt()) < 2e-5
    assert float((dw - wkf.grad).abs().max()) / float(wkf.grad.pow(2).mean().sqrt()) < 3e-4      # 295k pixel terms per weight in f32


@pytest.mark.parametrize("Cc,K", [(96, 96), (288, 96)])
def test_h2_full_size_96_channel_level(env, Cc, K):
    """C3's outer refiner level at full size (16 x 1024 x 144 pixels): conv_h2s96_kernel (forward, and the input gradient with 96 dy
    channels) against the f32-tensor bf16 engine of igemm.hip on the same bf16-representable operands - both have exact products
    and f32 accumulation - and an image copied to the end of the batch reproduces the first one (addressing of the last tile rows)."""
    ops = env
    torch.manual_seed(6)
    N, H, W = 16, 1024, 144
    x = torch.randn(N, H, W, Cc, device=DEV).to(BF16)
    x[N - 1] = x[0]
    w = (torch.randn(K, Cc, 3, 3, device=DEV) * 0.05).to(BF16).float()
    wk, wc = ops.weight_prep(w, None, True)
    d = ops.conv_desc(N, H, W, Cc, K, 3, 3, pad=(1, 1), precision="bf16")
    y = ops.conv_fprop_h2(d, x, ops.h2_weight_image(d, wk, False), None, out_dtype=torch.float32)
    assert ops.L().mq_last_kernel().decode().startswith("conv_h2s96")
    assert torch.equal(y[N - 1], y[0])
    y_old = ops.conv(x.float(), wk, wc, None, d)
    assert float((y - y_old).abs().max()) / float(y_old.pow(2).mean().sqrt()) < 2e-5
    del y, y_old
    r = torch.randn(N, H, W, K, device=DEV).to(BF16)
    r[N - 1] = r[0]
    dx = ops.conv_dgrad_h2(d, r, ops.h2_weight_image(d, wc, True), out_dtype=torch.float32)
    assert ops.L().mq_last_kernel().decode().startswith("conv_h2s96") == (K % 96 == 0 and K % 64 != 0)
    assert torch.equal(dx[N - 1], dx[0])
    xf = x.float().requires_grad_(True)
    wkf = wk.clone().requires_grad_(True)
    (ops.conv(xf, wkf, wc, None, d) * r.float()).sum().backward()
    assert float((dx - xf.grad).abs().max()) / float(xf.grad.pow(2).mean().sqrt()) < 2e-5
    # the weight gradient of the level: conv_wgrad_h2s96_kernel<1, 1> (K = 96: 96-row tiles, one-wrap walk over 146 padded columns,
    # hundreds of splits) against the f32-tensor engine's, 2.36 M pixel terms per weight; and its bias gradient against a plain sum
    dw, db = ops.conv_wgrad_h2(d, x, r, tuple(wk.shape), True)
    assert ops.L().mq_last_kernel().decode() == "conv_wgrad_h2s96_kernel<1, 1>"
    assert float((dw - wkf.grad).abs().max()) / float(wkf.grad.pow(2).mean().sqrt()) < 1e-3
    assert float((dw - wkf.grad).pow(2).sum().sqrt()) / float(wkf.grad.pow(2).sum().sqrt()) < 1e-4
    db_ref = r.float().sum(dim=(0, 1, 2), dtype=torch.float64)
    assert float((db.double() - db_ref).abs().max()) / float(db_ref.pow(2).mean().sqrt()) < 1e-4


def test_h2_random_shapes_against_float64():
    """tools/fuzz_h2.py: 40 random layers (channel counts, taps, strides, groups, dilation, ragged sizes drawn so that every kernel
    family of igemm_h2.hip gets its share) through fprop / dgrad / wgrad against float64: relative L2 <= 4e-5, no element further than
    4e-4 of the tensor's rms.  The tool exits non-zero at the first case above the bound."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_h2.py"), "40", "3"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "FUZZ OK 40 cases" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    assert "conv_h2s96q_kernel<3>" in out.stdout and "conv_h2s_kernel<" in out.stdout


MID_CFG = {
    "model": {"mel_channels": 32,
              "generator": {"channels": [32, 32, 64], "kernel_sizes": [3, 5], "fsq_levels": [8, 5, 5, 5],
                            "dropout": 0.1, "refiner_base_channels": 32, "refiner_depth": 3},
              "discriminator_patch": {"hidden_channels": [32, 32, 64, 64, 64], "kernel_sizes": [5, 5, 5, 3, 3, 3],
                                      "strides": [[1, 2], [2, 2], [2, 2], [2, 1], [2, 1], [2, 1]]},
              "discriminator_multibin": {"hidden_channels": [32, 32, 64, 64, 96], "kernel_sizes": [7, 5, 3, 3, 3, 3],
                                         "n_bins": 8, "n_no_strides": 2}},
    "training": {"lr": 1e-4, "beta1": 0.9, "beta2": 0.999, "lr_d_factor": 1.15, "d_beta1": 0.5, "d_beta2": 0.999,
                 "warmup_steps": 1000, "discriminator_train_start_epoch": 0,
                 "loss_weights": {"fm_lambda": 0.25, "Gloss_lambda": 15.0, "recon_lambda": 15.0},
                 "use_fm_loss": False, "seed": 42},
}


@pytest.mark.parametrize("use_fm", [False, True])
def test_bf16_storage_train_step_tracks_fp32(env, use_fm):
    """A model wide enough for the H2 kernels (all channel counts multiples of 32): two optimisation steps with
    training.precision = bf16 (bf16 activations in the refiner and the discriminators, H2 convolutions, fused
    LeakyReLU epilogues) against the same steps on the exact f32 engine: every loss within 3 %, gradient norms within
    6 %, and the H2 kernels must really have run."""
    ops = env
    from mqgan_amd.train import Trainer, synthetic_mel
    B, T, mel = 4, 128, 32
    x = torch.from_numpy(np.stack([synthetic_mel(T, mel, 40 + b) for b in range(B)])).to(DEV)
    lens = torch.tensor([T, 100, T, 77], dtype=torch.int32, device=DEV)
    res = {}
    for prec in ("fp32", "bf16"):
        cfg = {"model": MID_CFG["model"], "training": dict(MID_CFG["training"], use_fm_loss=use_fm, pretrained=None, precision=prec)}
        torch.manual_seed(7)
        tr = Trainer(cfg, device=DEV, build_data=False)
        tr.generator.eval()
        tr.epoch = 1
        timer = ops.KernelTimer()
        ops.TIMER = timer
        try:
            outs = []
            for _ in range(2):
                out = tr.train_step(x, lens)
                outs.append({k: float(v) for k, v in out.items()} | {"ng": float(tr.optimizer_g.grad_norm()), "nd": float(tr.optimizer_d.grad_norm())})
        finally:
            ops.TIMER = None
        torch.cuda.synchronize()
        res[prec] = (outs, set(timer.summary()))
    tags = res["bf16"][1]
    assert any("conv_h2_kernel" in t for t in tags) and any("conv_wgrad_h2_kernel" in t for t in tags), tags
    assert not any("h2" in t for t in res["fp32"][1])
    for s_ in range(2):
        a, b = res["bf16"][0][s_], res["fp32"][0][s_]
        for k in ("loss_d", "loss_g_total", "loss_recon_pre", "loss_recon_post", "loss_gan") + (("loss_fm",) if use_fm else ()):
            np.testing.assert_allclose(a[k], b[k], rtol=3e-2, atol=1e-4, err_msg=f"step {s_} {k}")
        np.testing.assert_allclose([a["ng"], a["nd"]], [b["ng"], b["nd"]], rtol=6e-2, err_msg=f"step {s_} norms")


def test_bf16_refiner_zero_padded_channel_storage(env, monkeypatch):
    """hifimusic's refiner has a 96-channel level (reference preencoder.py:134-202 with refiner_base_channels 96): under bf16
    storage its tensors CAN be kept with 128 channels, the last 32 zero (nn_base.pad_channels; opt-in, measured slower), so that
    the bf16-storage kernels walk 64-channel chunks of whole 128-byte lines.  The padded run must equal the unpadded one up to
    the bf16 rounding both carry (same products, another summation order), in the output and in EVERY parameter gradient; both
    must track the f32 refiner; and the padded run must really have run that level with 128 / 320 stored channels."""
    ops = env
    from mqgan_amd.preencoder import UNetRefiner
    torch.manual_seed(0)
    m = UNetRefiner(1, base_ch=96, depth=2, dropout=0.1, input_out_channels=[40, 32]).to(DEV).eval()
    B, T, Fq = 2, 64, 40
    x = torch.randn(B, T, Fq, 1, device=DEV)
    mask = torch.zeros(B, T, dtype=torch.uint8, device=DEV)
    mask[1, 45:] = 1
    r = torch.randn(B, T, 32, device=DEV)
    runs = {}
    for tag, storage, pad in (("f32", False, True), ("padded", True, True), ("unpadded", True, False)):
        monkeypatch.setattr(ops, "H2_PAD_CHANNELS", pad)
        for p in m.parameters():
            p.grad = None
        timer = ops.KernelTimer()
        ops.TIMER = timer
        try:
            with ops.conv_precision("bf16" if storage else "f32"), ops.request_act_storage(storage):
                out = m(x, mask)
                (out * r).sum().backward()
        finally:
            ops.TIMER = None
        torch.cuda.synchronize()
        runs[tag] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None},
                     set(timer.summary().keys()), list(timer.details))
    assert runs["padded"][0].shape == runs["f32"][0].shape == (B, T, 32)
    assert len(runs["padded"][1]) == len(runs["f32"][1]) > 20
    assert_rel(runs["padded"][0], runs["unpadded"][0], 1e-2, "output, padded vs unpadded storage")
    assert_rel(runs["padded"][0], runs["f32"][0], 3e-2, "output, padded bf16 vs f32")
    for k, gref in runs["f32"][1].items():
        if float(gref.norm()) == 0.0:
            continue
        assert tuple(runs["padded"][1][k].shape) == tuple(gref.shape), k              # gradients come back in the layer's own shape
        assert_rel(runs["padded"][1][k], runs["unpadded"][1][k], 3e-2, f"grad {k}, padded vs unpadded storage")
        assert_rel(runs["padded"][1][k], gref, 6e-2, f"grad {k}, padded bf16 vs f32")
    # the 96 / 288-channel reductions ran with 128 / 320 stored channels in the padded run (descriptor keys of the timed launches)
    chans = lambda run: {(k[3], k[4]) for k in run[3] if k is not None}               # (C, K) of every timed conv launch
    assert {(96, 96), (288, 96)} <= chans(runs["unpadded"]), chans(runs["unpadded"])
    assert {(128, 128), (320, 128)} <= chans(runs["padded"]) and not any(c in (96, 288) for c, _ in chans(runs["padded"])), chans(runs["padded"])
