"""Input pipeline for the trainer: the reference loads with ``num_workers: 0`` on the
training thread (train.py:262-269), which caps multi-GPU scaling once the step is fast.  PrefetchLoader
moves host work off the critical path: a background thread pulls collated batches from the DataLoader,
stages them in pinned memory and issues the host-to-device copies on a side HIP stream ``depth`` batches
ahead; the training stream only waits on the copy's event.

Round 6: the loader outlives an epoch.  ``epoch(e, next_epoch=e + 1)`` starts the producer of epoch e + 1 the moment epoch e's
last batch has been staged, so the first batch of the next epoch loads during the last optimisation steps of this one: the
start-of-epoch stall (r3 measured 146 ms: 8 ms per step of an 18-step epoch, eight times that share under a DistributedSampler's
eight-times-shorter epochs) is paid once per run, not once per epoch.
"""
import queue
import threading

import torch


class PrefetchLoader:
    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), depth
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        self._ahead = None          # (epoch, queue, stop, thread) of a producer started before its epoch was asked for

    def __len__(self):
        return len(self.loader)

    @staticmethod
    def _put(q, stop, item):
        """Blocking put that gives up when the consumer has gone away (returns False)."""
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def _producer(self, q, stop, epoch=None, done=None):
        put = lambda item: self._put(q, stop, item)
        try:
            sampler = getattr(self.loader, "sampler", None)
            if epoch is not None and hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)                 # DistributedSampler: the shuffle of THIS epoch, set before iter()
            for batch in self.loader:
                if stop.is_set():
                    return
                real, lens, names = batch
                if real is None:
                    if not put((None, None, names, None)):
                        return
                    continue
                if self.stream is None:
                    if not put((real, lens, names, None)):
                        return
                    continue
                real, lens = real.pin_memory(), lens.pin_memory()
                with torch.cuda.stream(self.stream):
                    d_real = real.to(self.device, non_blocking=True)
                    d_lens = lens.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                if not put((d_real, d_lens, names, ev)):
                    return
        except BaseException as e:          # surface loader errors on the consumer side
            put(e)
        finally:
            if done is not None:
                done.set()                  # the loader of this epoch is exhausted: everything it produced is in the queue
            put(StopIteration)

    def _start(self, epoch):
        q, stop, done = queue.Queue(maxsize=self.depth), threading.Event(), threading.Event()
        th = threading.Thread(target=self._producer, args=(q, stop, epoch, done), daemon=True)
        th.start()
        return (epoch, q, stop, th, done)

    @staticmethod
    def _release(run):
        """Stop a producer, drop what it staged (pinned buffers, device batches) and wait for it."""
        _, q, stop, th, _ = run
        stop.set()
        while True:
            try:
                q.get_nowait()
            except queue.Empty:
                break
        th.join()

    def close(self):
        if self._ahead is not None:
            self._release(self._ahead)
            self._ahead = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __iter__(self):
        return self.epoch(None)

    def epoch(self, epoch, next_epoch=None):
        """Batches of one epoch.  ``epoch`` is handed to the sampler's set_epoch (None: leave the sampler alone).  With
        ``next_epoch`` the producer of that epoch is started as soon as this epoch's last batch has been staged - its first
        ``depth`` batches are then waiting when ``epoch(next_epoch)`` is called."""
        run, self._ahead = self._ahead, None
        if run is not None and run[0] != epoch:           # a producer for another epoch than the one asked for: drop it
            self._release(run)
            run = None
        if run is None:
            run = self._start(epoch)
        _, q, stop, th, done = run
        try:
            total = len(self.loader)
        except TypeError:
            total = None
        got = 0
        try:
            while True:
                item = q.get()
                if item is StopIteration:
                    break
                if isinstance(item, BaseException):
                    raise item
                real, lens, names, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    real.record_stream(torch.cuda.current_stream(self.device))
                    lens.record_stream(torch.cuda.current_stream(self.device))
                got += 1
                if next_epoch is not None and self._ahead is None and (done.is_set() or (total is not None and got >= total)):
                    # this is the epoch's last batch, or the producer has staged everything already (what is left sits in the
                    # queue): the next epoch's producer starts now - after this epoch's loader is exhausted (a moment at most:
                    # it has produced its last batch), so sampler and dataset are never iterated by two producers at once
                    if done.wait(timeout=2.0):
                        self._ahead = self._start(next_epoch)
                yield real, lens, names
            if next_epoch is not None and self._ahead is None:
                self._ahead = self._start(next_epoch)
        finally:
            # the consumer may stop early (exception in the step, break, generator closed): release the producer,
            # drop what it staged (pinned buffers, device batches) and wait for it
            self._release(run)
