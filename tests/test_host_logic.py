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
