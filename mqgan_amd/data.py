"""Input pipeline for the trainer: the reference loads with ``num_workers: 0`` on the
training thread (train.py:262-269), which caps multi-GPU scaling once the step is fast.  PrefetchLoader
moves host work off the critical path: a background thread pulls collated batches from the DataLoader,
stages them in pinned memory and issues the host-to-device copies on a side HIP stream ``depth`` batches
ahead; the training stream only waits on the copy's event.
"""
import queue
import threading

import torch


class PrefetchLoader:
    def __init__(self, loader, device, depth: int = 2):
        self.loader, self.device, self.depth = loader, torch.device(device), depth
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

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

    def _producer(self, q, stop):
        put = lambda item: self._put(q, stop, item)
        try:
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
            put(StopIteration)

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        th = threading.Thread(target=self._producer, args=(q, stop), daemon=True)
        th.start()
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
                yield real, lens, names
        finally:
            # the consumer may stop early (exception in the step, break, generator closed): release the producer,
            # drop what it staged (pinned buffers, device batches) and wait for it
            stop.set()
            while True:
                try:
                    q.get_nowait()
                except queue.Empty:
                    break
            th.join()
