"""Host-side logic that needs no GPU: the prefetching loader's life cycle and the collate function."""
import threading
import time

import numpy as np
import torch

from mqgan_amd.data import PrefetchLoader
from mqgan_amd.train import pad_collate_fn


class _Loader:
    def __init__(self, n):
        self.n = n
        self.produced = 0

    def __len__(self):
        return self.n

    def __iter__(self):
        for i in range(self.n):
            self.produced += 1
            yield torch.full((2, 4, 3), float(i)), torch.tensor([4, 3], dtype=torch.int32), ("a", "b")


def test_prefetch_loader_yields_everything_in_order():
    pl = PrefetchLoader(_Loader(7), "cpu", depth=2)
    got = [int(x[0, 0, 0]) for x, _, _ in pl]
    assert got == list(range(7))


def test_prefetch_loader_releases_its_thread_when_the_consumer_stops_early():
    """a consumer that breaks out (or raises) must not leave the producer blocked on the bounded queue."""
    before = threading.active_count()
    src = _Loader(1000)
    for i, (x, lens, names) in enumerate(PrefetchLoader(src, "cpu", depth=2)):
        if i == 3:
            break
    deadline = time.time() + 5.0
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() == before, "producer thread still alive"
    assert src.produced < 20, "producer ran on after the consumer left"

    class Boom(Exception):
        pass

    try:
        for i, _ in enumerate(PrefetchLoader(_Loader(1000), "cpu", depth=2)):
            if i == 2:
                raise Boom()
    except Boom:
        pass
    deadline = time.time() + 5.0
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() == before


def test_prefetch_loader_surfaces_loader_errors():
    class Bad:
        def __iter__(self):
            yield torch.zeros(1, 2, 3), torch.tensor([2], dtype=torch.int32), ("x",)
            raise ValueError("disk on fire")

    it = iter(PrefetchLoader(Bad(), "cpu", depth=2))
    next(it)
    try:
        next(it)
        assert False, "expected the loader's error"
    except ValueError as e:
        assert "disk on fire" in str(e)


def test_pad_collate_drops_failed_loads_and_pads_to_the_batch_crop():
    """train.py:140-197: None items are dropped, one crop length per batch, zero right-padding, int32 lengths."""
    items = [(torch.ones(50, 4), 50, "a"), None, (torch.ones(20, 4), 20, "b")]
    x, lens, names = pad_collate_fn(items, crop_lens=32)
    assert x.shape == (2, 32, 4) and lens.dtype == torch.int32 and lens.tolist() == [32, 20] and names == ("a", "b")
    assert float(x[1, 20:].abs().sum()) == 0.0
    assert pad_collate_fn([None, None], crop_lens=32) == (None, None, None)
    x, lens, _ = pad_collate_fn(items, crop_lens=None)
    assert x.shape == (2, 50, 4) and lens.tolist() == [50, 20]


def test_winograd_f2x2_3x3_identities():
    """The algebra csrc/winograd.hip implements, in float64: forward Y = A^T[(G g G^T) . (B^T d B)]A equals the direct 3x3
    correlation of a 4x4 tile; the weight gradient is G^T[(B^T d B) . (A dY A^T)]G; the input gradient is the same forward
    form applied to dy with the filter rotated by 180 degrees (a 'same' convolution)."""
    rng = np.random.default_rng(3)
    Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
    At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    d, g, dY = rng.standard_normal((4, 4)), rng.standard_normal((3, 3)), rng.standard_normal((2, 2))
    direct = np.array([[(d[a:a + 3, b:b + 3] * g).sum() for b in range(2)] for a in range(2)])
    V, U = Bt @ d @ Bt.T, G @ g @ G.T
    np.testing.assert_allclose(At @ (U * V) @ At.T, direct, rtol=1e-12, atol=1e-12)
    # d(sum(Y * dY)) / dg
    want_dg = np.array([[(d[r:r + 2, s:s + 2] * dY).sum() for s in range(3)] for r in range(3)])
    Z = At.T @ dY @ At
    np.testing.assert_allclose(G.T @ (V * Z) @ G, want_dg, rtol=1e-12, atol=1e-12)
    # input gradient of a padded 'same' convolution = 'same' convolution of dy with the rotated filter
    H = W = 6
    x, w, gy = rng.standard_normal((H, W)), rng.standard_normal((3, 3)), rng.standard_normal((H, W))
    xp = np.pad(x, 1)
    dx = np.zeros_like(xp)
    for h in range(H):
        for ww in range(W):
            dx[h:h + 3, ww:ww + 3] += gy[h, ww] * w
    gp = np.pad(gy, 1)
    rot = w[::-1, ::-1]
    same = np.array([[(gp[h:h + 3, ww:ww + 3] * rot).sum() for ww in range(W)] for h in range(H)])
    np.testing.assert_allclose(dx[1:-1, 1:-1], same, rtol=1e-12, atol=1e-12)


def test_bench_traffic_lookup_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.pmc_traffic: bytes per launch only from a PMC pass taken on the kernel sources that are being run (hash of the
    conv kernel files); another build reads as None with the file named as stale; unknown kernels read as None."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    bid = bench.kernel_build_id()
    assert len(bid) == 16 and bid == bench.kernel_build_id()
    (tmp_path / "profiles").mkdir()
    f = tmp_path / "profiles" / "r9_pmc_traffic.json"
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_build_id", lambda: bid)
    f.write_text(json.dumps({"_build": bid, "k<1>": {"fetch_bytes_per_launch": 10.0, "write_bytes_per_launch": 5.0, "launches": 3}}))
    assert bench.pmc_traffic("k<1>") == (15.0, os.path.join("profiles", "r9_pmc_traffic.json"))
    assert bench.pmc_traffic("other") == (None, None)
    f.write_text(json.dumps({"_build": "0" * 16, "k<1>": {"fetch_bytes_per_launch": 10.0, "write_bytes_per_launch": 5.0, "launches": 3}}))
    t, src = bench.pmc_traffic("k<1>")
    assert t is None and "stale" in src
    assert bench.peak_of("conv_h2_kernel<3, 2, 64, false, true>") == bench.PEAK_BF16_MFMA_TFLOPS
    assert bench.peak_of("conv_nt_kernel<128, false, 3>") == bench.PEAK_F32_MFMA_TFLOPS


def test_gather_table_is_plain_host_data_per_call():
    """FlatArena._gather_table: the rows handed to mq_multi_gather_args are built from scratch on every call (no staging
    buffer survives a call, so nothing can be overwritten while a launch is in flight - the r2 pinned-ring race), one row per
    parameter with a source (0 = zeros), skipping gradients that already alias the arena."""
    import torch
    from mqgan_amd.ddp import FlatArena

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Parameter(torch.randn(5, 3))
            self.s = torch.nn.Parameter(torch.tensor(2.0))
            self.b = torch.nn.Parameter(torch.randn(9))
    net = Net()
    arena = FlatArena([net], gather=True)
    assert not hasattr(arena, "_ring")
    ga, gb = torch.randn(5, 3), torch.randn(9)
    net.a.grad, net.b.grad = ga, gb
    t1 = arena._gather_table([0, 1, 2])
    assert t1 == [(ga.data_ptr(), 0, 15, 0), (0, 16, 1, 0), (gb.data_ptr(), 20, 9, 0)]
    ga2 = torch.randn(5, 3)
    net.a.grad = ga2
    t2 = arena._gather_table([0, 1, 2])
    assert t2[0][0] == ga2.data_ptr() and t1[0][0] == ga.data_ptr() and t1 is not t2      # the first table is untouched
    net.b.grad = arena.flat_grad[20:29]                         # already an arena view: nothing to move
    assert [r[1] for r in arena._gather_table([0, 1, 2])] == [0, 16]
    nc = torch.randn(3, 5).t()
    net.a.grad = nc                                             # made contiguous, and the copy is kept alive on the parameter
    row = arena._gather_table([0])[0]
    assert net.a.grad.is_contiguous() and row[0] == net.a.grad.data_ptr() and torch.equal(net.a.grad, nc)


def test_comm_active_flag():
    from mqgan_amd.comm import Comm

    class Two(Comm):
        world = 2
    assert not Comm().active and Two().active


def test_reencode_default_batching_is_the_reference_loop():
    """reencode_spectrograms.py:36-47 (reference): os.walk order cut into consecutive batches.  That is the DEFAULT policy of
    the drop-in entry points (a file's output depends on its batch neighbours, so another default would write other files
    than the reference for the same tree); 'sorted' / 'exact' are opt-in."""
    import inspect
    from mqgan_amd import reencode_spectrograms as R
    files = [f"f{i}.npy" for i in range(11)]
    lengths = [50, 7, 31, 50, None, 12, 90, 7, 31, 64, 5]
    ref = [files[i:i + 4] for i in range(0, len(files), 4)]
    assert R.plan_batches(files, lengths, 4) == ref
    assert R.plan_batches(files, lengths, 4, "order") == ref
    for fn in (R.plan_batches, R._run, R.reencode_spectrograms, R.reencode_spectrograms_from_checkpoint):
        assert inspect.signature(fn).parameters["bucket"].default == "order", fn.__name__
    srt = R.plan_batches(files, lengths, 4, "sorted", 0.25)
    assert srt != ref and sorted(f for b in srt for f in b) == sorted(files)


def test_winograd_form_selection_per_layer():
    """Which exact-f32 form ops picks for the convolutions of BASELINE config 2 (pure host logic + the library's tile tables):
    refiner 3x3 -> F(4x5) on T x 90 images, polyphase 5x5/s2 -> F(4x5) on 40 x 128 and 20 x 64, multi-bin (3 x k) ->
    F(5, 3) along 10 bins (nested along time when the column taps have stride 1), the patch discriminator's stride-(2, 1) layers
    and the 1-D convolutions -> F(4, 3) / F(4, 5) / F(2, 7) along time; small layers and bf16 stay on the direct kernels."""
    from mqgan_amd import ops
    L = ops.L()
    assert [L.mq_wino1d_tile(h) for h in (2, 6, 8, 9, 10, 16)] == [2, 2, 4, 5, 5, 4]
    assert [L.mq_winow_tile(s) for s in (3, 5, 7, 4)] == [4, 4, 2, 0]
    assert ops._wino_tile(512, 90) == (4, 5) and ops._wino_tile(64, 90) == (4, 5) and ops._wino_tile(40, 128) == (4, 5)
    assert ops._wino_tile(20, 64) == (4, 5) and ops._wino_tile(16, 16) == (4, 4) and ops._wino_tile(9, 9) == (3, 3)
    f32 = "f32"
    refiner = ops.conv_desc(32, 256, 90, 128, 128, 3, 3, pad=(1, 1), precision=f32)
    assert ops._wino_ok(refiner) and ops._wino_np(refiner) == 42 and ops._wino_tiles(refiner) == 32 * 64 * 18
    assert ops._wino_ok(ops.conv_desc(32, 512, 90, 64, 64, 3, 3, pad=(1, 1), precision=f32))           # C K / (C + K) = 32
    assert not ops._wino_ok(ops.conv_desc(32, 512, 90, 48, 64, 3, 3, pad=(1, 1), precision=f32))
    assert not ops._wino_ok(ops.conv_desc(32, 256, 90, 128, 128, 3, 3, pad=(1, 1), precision="bf16"))
    # the larger tiles' fused dy pass takes K <= 1024: wider layers stay on the direct kernels instead of failing in backward
    assert ops._wino_ok(ops.conv_desc(2, 64, 90, 1024, 1024, 3, 3, pad=(1, 1), precision=f32))
    assert not ops._wino_ok(ops.conv_desc(2, 64, 90, 512, 1040, 3, 3, pad=(1, 1), precision=f32))
    patch = ops.conv_desc(64, 80, 256, 256, 256, 5, 5, stride=(2, 2), pad=(2, 2), precision=f32)
    assert ops._wino_s2d_ok(patch) and ops._s2d_tile(patch) == (4, 5)
    mb35 = ops.conv_desc(512, 10, 512, 128, 128, 3, 5, pad=(1, 2), groups=8, precision=f32)
    mb33 = ops.conv_desc(512, 10, 512, 128, 256, 3, 3, stride=(1, 2), pad=(1, 1), groups=8, precision=f32)
    assert ops._wino1d_ok(mb35) and ops._wino1d_ok(mb33) and ops._wino1d_m(mb35) == 5
    inner35, inner33 = ops._wino1d_gemm_desc(mb35, 5), ops._wino1d_gemm_desc(mb33, 5)
    assert (inner35.N, inner35.H, inner35.S, inner35.groups) == (7 * 512 * 2, 1, 5, 56)
    assert ops._winow_ok(inner35) and not ops._winow_ok(inner33)                     # stride 2 along time: no column form
    g = ops._ww_desc(inner35)
    assert (g.N, g.W, g.R, g.S, g.groups) == (8 * inner35.N, 128, 1, 1, 8 * 56)
    d1 = ops.conv_desc(64, 20, 64, 384, 512, 3, 3, stride=(2, 1), pad=(1, 1), precision=f32)
    assert not ops._wino_ok(d1) and ops._winow_ok(d1) and ops._ww_geom(d1) == (4, 16, 6)
    dec = ops.conv_desc(32, 1, 512, 768, 512, 1, 7, pad=(0, 6), out_hw=(1, 512), precision=f32)       # causal k = 7
    assert ops._winow_ok(dec) and ops._ww_geom(dec) == (2, 256, 8)
    assert not ops._winow_ok(ops.conv_desc(32, 1, 512, 768, 512, 1, 7, pad=(0, 6), dil=(1, 2), out_hw=(1, 512), precision=f32))
