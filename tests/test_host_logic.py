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


def test_prefetch_loader_stages_the_next_epoch_during_the_last_steps_of_this_one():
    """Round 6: the loader lives across epochs.  epoch(e, next_epoch=e + 1) starts the producer of epoch
    e + 1 when epoch e's last batch has been staged: while the consumer is still inside the last step of epoch e the first
    `depth` batches of e + 1 are already produced; the sampler's set_epoch is called once per epoch, in order, by the producer
    of that epoch, never while another epoch is being iterated; asking for another epoch than the staged one drops it; close()
    leaves no thread behind."""
    before = threading.active_count()

    class Sampler:
        def __init__(self):
            self.calls = []

        def set_epoch(self, e):
            self.calls.append(e)

    class Epochs:
        def __init__(self, n):
            self.n, self.sampler, self.active, self.iterated, self.max_active = n, Sampler(), 0, 0, 0

        def __len__(self):
            return self.n

        def __iter__(self):
            e = self.sampler.calls[-1] if self.sampler.calls else -1
            self.active += 1
            self.max_active = max(self.max_active, self.active)
            self.iterated += 1
            try:
                for i in range(self.n):
                    yield torch.full((1, 2, 3), float(100 * e + i)), torch.tensor([2], dtype=torch.int32), ("f",)
            finally:
                self.active -= 1

    src = Epochs(5)
    pl = PrefetchLoader(src, "cpu", depth=2)
    seen = []
    for e in (1, 2, 3):
        for i, (x, _, _) in enumerate(pl.epoch(e, next_epoch=e + 1 if e < 3 else None)):
            seen.append(int(x[0, 0, 0]))
            if i == 4 and e < 3:
                # inside the LAST step of epoch e: give the next producer a moment, then look - it must have staged batches already
                deadline = time.time() + 5.0
                while src.iterated < e + 1 and time.time() < deadline:
                    time.sleep(0.01)
                assert src.iterated == e + 1, "the next epoch's producer did not start during this epoch's last step"
    assert seen == [100 * e + i for e in (1, 2, 3) for i in range(5)]
    assert src.sampler.calls == [1, 2, 3] and src.max_active == 1
    # a staged epoch nobody asks for: replaced by the one asked for, and close() joins whatever is left
    for _ in pl.epoch(7, next_epoch=8):
        pass
    assert [int(x[0, 0, 0]) for x, _, _ in pl.epoch(9)][0] == 900          # epoch 8 was staged, 9 asked for
    for _ in pl.epoch(10, next_epoch=11):
        pass
    pl.close()
    deadline = time.time() + 5.0
    while threading.active_count() > before and time.time() < deadline:
        time.sleep(0.05)
    assert threading.active_count() == before, "a producer thread outlived close()"


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
    for name in ("conv_h2s_kernel<3, 3, false>", "conv_h2s96q_kernel<3>", "conv_h2s96_kernel<3>", "conv_h2_small_kernel<2, 32>",
                 "conv_wgrad_h2s_kernel<1>", "conv_wgrad_h2s96_kernel<1, 1>"):
        assert bench.peak_of(name) == bench.PEAK_BF16_MFMA_TFLOPS, name
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


def test_conv_plan_form_selection_per_layer():
    """mq_conv_plan (csrc/plan.cpp: the ONE copy of the eligibility logic; host code, no GPU needed): which exact-f32 form the
    convolutions of BASELINE config 2 take - refiner 3x3 -> F(4x5) on T x 90 images, polyphase 5x5/s2 -> F(4x5) on 40 x 128 and
    20 x 64, multi-bin (3 x k) -> F(5, 3) along 10 bins (nested along time when the column taps have stride 1), the patch
    discriminator's stride-(2, 1) layers and the 1-D convolutions -> F(4, 3) / F(4, 5) / F(2, 7) along time; small layers and bf16
    stay on the direct kernels - and the GEMM descriptors / buffer sizes it reports."""
    from mqgan_amd import ops
    L = ops.L()
    assert [L.mq_wino1d_tile(h) for h in (2, 6, 8, 9, 10, 16)] == [2, 2, 4, 5, 5, 4]
    assert [L.mq_winow_tile(s) for s in (3, 5, 7, 4)] == [4, 4, 2, 0]
    f32 = "f32"

    def tile(H, W):
        p = ops.conv_plan(ops.conv_desc(1, H, W, 128, 128, 3, 3, pad=(1, 1), precision=f32), False)
        assert p.form == ops.FORM_WINO2D
        return p.mh, p.mw
    assert tile(512, 90) == (4, 5) and tile(64, 90) == (4, 5) and tile(40, 128) == (4, 5)
    assert tile(20, 64) == (4, 5) and tile(16, 16) == (4, 4) and tile(9, 9) == (3, 3)
    refiner = ops.conv_desc(32, 256, 90, 128, 128, 3, 3, pad=(1, 1), precision=f32)
    p = ops.conv_plan(refiner, False)
    assert ops.form_name(p) == "winograd2d" and p.npos == 42 and p.tiles == 32 * 64 * 18
    assert (p.gemm.N, p.gemm.H, p.gemm.W, p.gemm.R, p.gemm.S, p.gemm.groups) == (42 * 32 * 64, 1, 18, 1, 1, 42)
    assert p.flops == 2.0 * 42 * p.tiles * 128 * 128
    assert p.keep_bytes >= 4 * 42 * p.tiles * 128 and p.ws_fprop >= 4 * 42 * p.tiles * 256 and p.ws_backward > p.ws_fprop
    assert ops.conv_plan(refiner, True).form != ops.FORM_WINO2D            # a LeakyReLU 3x3 layer goes the row way (F(m, 3))
    assert ops._wino_ok(ops.conv_desc(32, 512, 90, 64, 64, 3, 3, pad=(1, 1), precision=f32))           # C K / (C + K) = 32
    assert not ops._wino_ok(ops.conv_desc(32, 512, 90, 48, 64, 3, 3, pad=(1, 1), precision=f32))
    assert not ops._wino_ok(ops.conv_desc(32, 256, 90, 128, 128, 3, 3, pad=(1, 1), precision="bf16"))
    # the fused dy pass takes K <= 1024: wider layers stay on the direct kernels instead of failing in backward
    assert ops._wino_ok(ops.conv_desc(2, 64, 90, 1024, 1024, 3, 3, pad=(1, 1), precision=f32))
    assert not ops._wino_ok(ops.conv_desc(2, 64, 90, 512, 1040, 3, 3, pad=(1, 1), precision=f32))
    patch = ops.conv_desc(64, 80, 256, 256, 256, 5, 5, stride=(2, 2), pad=(2, 2), precision=f32)
    p = ops.conv_plan(patch, True)
    assert p.form == ops.FORM_S2D and (p.mh, p.mw) == (4, 5) and p.gemm.C == 4 * 256 and p.gemm.groups == 42
    assert ops.conv_plan(patch, False).form == ops.FORM_DIRECT              # the polyphase form exists for the leaky layers
    mb35 = ops.conv_desc(512, 10, 512, 128, 128, 3, 5, pad=(1, 2), groups=8, precision=f32)
    mb33 = ops.conv_desc(512, 10, 512, 128, 256, 3, 3, stride=(1, 2), pad=(1, 1), groups=8, precision=f32)
    p35, p33 = ops.conv_plan(mb35, True), ops.conv_plan(mb33, True)
    assert p35.form == ops.FORM_WINO1D and p33.form == ops.FORM_WINO1D and p35.mh == 5
    assert (p35.gemm.N, p35.gemm.H, p35.gemm.S, p35.gemm.groups) == (7 * 512 * 2, 1, 5, 56)
    assert p35.nested == 1 and p33.nested == 0                              # stride 2 along time: no column form inside
    g = p35.inner
    assert (g.N, g.W, g.R, g.S, g.groups) == (8 * p35.gemm.N, 128, 1, 1, 8 * 56)
    d1 = ops.conv_desc(64, 20, 64, 384, 512, 3, 3, stride=(2, 1), pad=(1, 1), precision=f32)
    p = ops.conv_plan(d1, True)
    assert not ops._wino_ok(d1) and p.form == ops.FORM_WINOW and (p.mw, p.npos) == (4, 6) and p.gemm.W == 16
    dec = ops.conv_desc(32, 1, 512, 768, 512, 1, 7, pad=(0, 6), out_hw=(1, 512), precision=f32)       # causal k = 7
    p = ops.conv_plan(dec, False)
    assert p.form == ops.FORM_WINOW and (p.mw, p.npos, p.gemm.W) == (2, 8, 256)
    assert not ops._winow_ok(ops.conv_desc(32, 1, 512, 768, 512, 1, 7, pad=(0, 6), dil=(1, 2), out_hw=(1, 512), precision=f32))
    # the switches are values of the plan's configuration
    import pytest
    keep = ops.WINO_TILE
    try:
        ops.WINO_TILE = "3"
        assert tile(512, 90) == (3, 3)
        ops.WINO_TILE = "6"                      # built, never chosen unless asked for
        assert tile(512, 90) == (6, 6)
        ops.WINO_TILE = "2"
        with pytest.raises(Exception, match="auto, 3, 4, 45 or 6"):
            tile(512, 90)
    finally:
        ops.WINO_TILE = keep


def test_conv_plan_of_every_config2_layer():
    """The plan of every convolution of BASELINE config 2 (hifispeech, batch 32 x 512 frames x 80 mels) that the default path
    runs in a reduced-multiplication form: the table DESIGN 3.6-3.9 describes, as the library computes it."""
    from mqgan_amd import ops
    f32 = "f32"
    W2 = "winograd2d"
    table = []
    # refiner (preencoder.py:86-130): T x 90 images at 512 / 256 / 128 / 64 frames, 64 ... 512 channels
    for T, cin, cout in ((512, 64, 64), (256, 64, 128), (256, 128, 128), (128, 128, 256), (128, 256, 256), (64, 256, 512), (64, 512, 512),
                         (128, 768, 256), (256, 384, 128), (512, 192, 64)):
        table.append((ops.conv_desc(32, T, 90, cin, cout, 3, 3, pad=(1, 1), precision=f32), False, W2, (4, 5)))
    # patch discriminator (discriminators.py:196-242), real + fake as one batch of 64 in eval mode
    table.append((ops.conv_desc(64, 80, 256, 256, 256, 5, 5, stride=(2, 2), pad=(2, 2), precision=f32), True, "polyphase-winograd2d", (4, 5)))
    table.append((ops.conv_desc(64, 40, 128, 256, 384, 5, 5, stride=(2, 2), pad=(2, 2), precision=f32), True, "polyphase-winograd2d", (4, 5)))
    table.append((ops.conv_desc(64, 20, 64, 384, 512, 3, 3, stride=(2, 1), pad=(1, 1), precision=f32), True, "winograd-columns", (0, 4)))
    table.append((ops.conv_desc(64, 10, 64, 512, 512, 3, 3, stride=(2, 1), pad=(1, 1), precision=f32), True, "winograd-columns", (0, 4)))
    # multi-bin discriminator (discriminators.py:292-311): 8 bands x 64 images of 10 bins
    table.append((ops.conv_desc(512, 10, 512, 128, 128, 3, 5, pad=(1, 2), groups=8, precision=f32), True, "winograd-rows", (5, 4)))
    table.append((ops.conv_desc(512, 10, 512, 128, 256, 3, 3, stride=(1, 2), pad=(1, 1), groups=8, precision=f32), True, "winograd-rows", (5, 0)))
    table.append((ops.conv_desc(512, 10, 256, 256, 256, 3, 3, stride=(1, 2), pad=(1, 1), groups=8, precision=f32), True, "winograd-rows", (5, 0)))
    table.append((ops.conv_desc(512, 10, 128, 256, 384, 3, 3, stride=(1, 2), pad=(1, 1), groups=8, precision=f32), True, "winograd-rows", (5, 0)))
    # encoder / decoder 1-D convolutions (attentions.py:95-130): k = 3, 3, 5, 7; causal left padding in the decoder
    for cin, cout, k in ((512, 512, 3), (512, 512, 5), (512, 768, 7), (768, 768, 7)):
        table.append((ops.conv_desc(32, 1, 512, cin, cout, 1, k, pad=(0, k // 2), precision=f32), False, "winograd-columns", (0, 2 if k == 7 else 4)))
        table.append((ops.conv_desc(32, 1, 512, cout, cin, 1, k, pad=(0, k - 1), out_hw=(1, 512), precision=f32), False, "winograd-columns",
                      (0, 2 if k == 7 else 4)))
    # what stays direct: single-channel first layers / logits (thin path in Python), the 12-channel entry of the refiner
    table.append((ops.conv_desc(32, 512, 90, 64, 12, 1, 1, precision=f32), False, "direct", (0, 0)))
    table.append((ops.conv_desc(64, 80, 512, 1, 256, 5, 5, stride=(1, 2), pad=(2, 2), precision=f32), True, "direct", (0, 0)))
    for d, leaky, form, tile in table:
        p = ops.conv_plan(d, leaky)
        assert ops.form_name(p) == form and (p.mh, p.mw) == tile, (ops._desc_key(d), leaky, ops.form_name(p), p.mh, p.mw)
        if form != "direct":
            direct = 2.0 * d.N * d.P * d.Q * d.K * d.C * d.R * d.S
            assert p.flops < 0.62 * direct and p.ws_fprop > 0 and p.ws_backward > 0 and p.keep_bytes > 0

