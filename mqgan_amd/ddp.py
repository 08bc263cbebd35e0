"""Data-parallel gradient exchange for one process per GPU (torch.distributed: backend "nccl" is
RCCL over xGMI on ROCm; "gloo" on CPU for tests).

Design (MI355X-first, not a port of torch DDP):
  * parameters and gradients of a model live in two flat fp32 arenas (FlatArena), so the optimiser,
    the gradient norm and the all-reduce all work on a handful of large contiguous ranges;
  * GradReducer cuts the gradient arena into a few large buckets (default 32 MiB: a ring all-reduce
    over xGMI is per-link bound, few large messages beat many small ones) in REVERSE registration
    order, launches an asynchronous all-reduce for a bucket as soon as every gradient in it has been
    accumulated (post-accumulate-grad hooks, i.e. while backward is still running; the collective
    runs on the process group's own HIP stream), and waits for all of them before the clip/Adam
    kernels read the arena;
  * the buckets TAPER towards the front of the arena (r5): the gradients of the first-registered
    layers become ready when backward ends, so whatever is still queued on the collective stream at
    that moment is exposed - no compute is left to hide it.  A bucket therefore takes at most half
    of what remains in front of it (never more than ``bucket_bytes``, never less than
    ``tail_bytes`` = 1 MiB): ..., 32, 16, 8, 4, 2, 1 MiB.  The last-ready bucket is the smallest,
    and the ones launched in the last stretch of backward shrink with the time left to hide them
    (r4 cut equal 32 MiB buckets and left the remainder - 9.4 MiB for the generator - to the end);
  * parameters that never receive a gradient (``hidden_proj.*``: its only consumer is detached,
    reference preencoder.py:413-415) do not block a bucket: whatever is still pending is flushed by
    ``finish()``.
"""
from typing import Iterable, List

import ctypes as C
import os

import torch
import torch.distributed as dist

_TRACE = os.environ.get("MQGAN_DDP_TRACE") == "1"      # one stderr line per bucket all-reduce (which rank launches what, when)
_BLOCKING = os.environ.get("MQGAN_DDP_BLOCKING") == "1"  # debugging: blocking bucket all-reduces instead of asynchronous ones
_ASYNC = os.environ.get("MQGAN_DDP_ASYNC") == "1"        # debugging: no host drain in front of gloo device collectives (reproduces the r2 hang)


class FlatArena:
    """Re-homes the parameters of ``modules`` into one flat buffer and gives every parameter a
    gradient view into a second flat buffer (zeroed, never None).

    ``gather=True`` (the trainer's mode): ``.grad`` is None between steps, autograd leaves each gradient in the tensor its
    backward kernel wrote (no per-parameter accumulation launch into a zeroed view), and ``gather()`` fills the gradient arena
    from all of them with ONE kernel (zeros where a parameter got no gradient) - per bucket when a GradReducer drives it."""

    MG_CHUNK = 2048                # elements per block of mq_multi_gather

    def __init__(self, modules: Iterable[torch.nn.Module], align: int = 4, gather: bool = False):
        self.params: List[torch.nn.Parameter] = []
        seen = set()
        for m in modules:
            for p in m.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    self.params.append(p)
        if not self.params:
            raise ValueError("FlatArena: no parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + align - 1) // align * align          # keep every tensor 16-byte aligned
        self.numel = n
        self.flat_param = torch.zeros(n, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(n, dtype=dt, device=dev)
        self.gather_mode = bool(gather)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat_param[o:o + p.numel()].view_as(p)
                view.copy_(p)
                p.data = view
                p.grad = None if self.gather_mode else self.flat_grad[o:o + p.numel()].view_as(p)
        self._gathered = [False] * len(self.params)

    def zero_grad(self):
        if self.gather_mode:
            for p in self.params:
                p.grad = None
            self._gathered = [False] * len(self.params)
            return
        self.flat_grad.zero_()
        for p, o in zip(self.params, self.offsets):               # re-attach if something reset .grad to None
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                p.grad = self.flat_grad[o:o + p.numel()].view_as(p)

    def gather(self, indices=None):
        """Gradient arena <- the parameters' gradient tensors (``indices``: positions in ``params``, default all that were
        not gathered since the last zero_grad).  No-op outside gather mode."""
        if not self.gather_mode:
            return
        todo = [i for i in (range(len(self.params)) if indices is None else indices) if not self._gathered[i]]
        if not todo:
            return
        for i in todo:
            self._gathered[i] = True
        if self.flat_grad.is_cuda:
            self._gather_hip(todo)
        else:                      # CPU arenas exist for the gloo tests of the reducer logic only
            with torch.no_grad():
                for i in todo:
                    p, o = self.params[i], self.offsets[i]
                    if p.grad is None:
                        self.flat_grad[o:o + p.numel()].zero_()
                    else:
                        self.flat_grad[o:o + p.numel()].copy_(p.grad.reshape(-1))

    def _gather_table(self, todo):
        """Host rows {source address or 0, arena offset, elements, 0} of the gradients to move (pure host logic)."""
        from . import _lib
        base = self.flat_grad.data_ptr()
        table = []
        for i in todo:
            p, o = self.params[i], self.offsets[i]
            g = p.grad
            src = 0
            if g is not None:
                if g.dtype != self.flat_grad.dtype or g.device != self.flat_grad.device:
                    raise _lib.MqganHipError(f"FlatArena.gather: gradient of parameter {i} is {g.dtype} on {g.device}")
                if not g.is_contiguous():
                    g = g.contiguous()
                    p.grad = g                                  # keeps the copy alive until the kernel has run
                src = g.data_ptr()
            if src == base + 4 * o:                             # already an arena view (legacy holders): nothing to move
                continue
            table.append((src, o, p.numel(), 0))
        return table

    def _gather_hip(self, todo):
        # The rows go to the kernel BY VALUE (mq_multi_gather_args copies them into the launch's arguments before it
        # returns), so nothing the host does later - however many steps it runs ahead of the stream, whatever addresses
        # the allocator hands out for the next step's gradients - can change what an enqueued launch reads.  (r2 staged
        # the table in a pinned ring that the host could wrap before the copy had executed.)
        from .ops import L, check, ptr, stream
        table = self._gather_table(todo)
        if not table:
            return
        rows = (C.c_int64 * (4 * len(table)))(*[v for r in table for v in r])
        check(L().mq_multi_gather_args(rows, len(table), ptr(self.flat_grad), stream()), "mq_multi_gather_args")

    def rebind(self):
        """After load_state_dict / .to(): make sure parameters still alias the arena."""
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                if p.data.data_ptr() != self.flat_param.data_ptr() + 4 * o:
                    view = self.flat_param[o:o + p.numel()].view_as(p)
                    view.copy_(p.data)
                    p.data = view


class GradReducer:
    def __init__(self, arena: FlatArena, process_group=None, bucket_bytes: int = 32 << 20, average: bool = False, comm=None,
                 force: bool = False, tail_bytes: int = 1 << 20):
        """``force``: hooks, per-bucket arena fill and asynchronous all-reduces even in a process group of ONE rank (the sum
        is then the identity) - lets a one-GPU box execute the production RCCL path end to end."""
        self.arena = arena
        self.pg = process_group
        # ``comm``: a non-torch.distributed communicator (mqgan_amd.comm.Comm with world > 1, e.g. a test double): the whole
        # gradient arena is then summed in finish() with one blocking call instead of hooked, bucketed async all-reduces
        # ``comm.buckets_async`` (comm.RcclAbiComm): the buckets go through the communicator's own asynchronous all-reduce - the
        # C ABI's mq_allreduce_bucket on a side stream - from the same hooks, instead of torch.distributed's
        self._abi = comm if getattr(comm, "buckets_async", False) else None
        self.comm = comm if (comm is not None and comm.world > 1 and not dist.is_initialized() and self._abi is None) else None
        self.world = (self._abi.world if self._abi is not None else self.comm.world if self.comm is not None
                      else (dist.get_world_size(process_group) if dist.is_initialized() else 1))
        self._rank = self._abi.rank if self._abi is not None else (dist.get_rank(process_group) if dist.is_initialized() else 0)
        self.average = average
        self.buckets = []                      # (start, end) element ranges of flat_grad, reverse order
        self.bucket_members = []               # positions in arena.params, per bucket
        self.param_bucket = {}
        cap = max(1, bucket_bytes // 4)
        floor = max(1, min(cap, tail_bytes // 4))
        end = arena.numel
        cur_start, members = end, []
        index_of = {id(p): i for i, p in enumerate(arena.params)}
        self._index_of = index_of
        # tapering: a bucket that ends at `end` closes once it holds min(cap, max(floor, end / 2)) elements - half of what is
        # left in front of it (the arena starts at 0), so the bucket sizes fall geometrically towards the last-ready gradients
        want = min(cap, max(floor, end // 2))
        for p, o in reversed(list(zip(arena.params, arena.offsets))):
            cur_start = o
            members.append(p)
            if end - cur_start >= want:
                self._add_bucket(cur_start, end, members)
                end, members = cur_start, []
                want = min(cap, max(floor, end // 2))
        if members:
            self._add_bucket(cur_start, end, members)
        self._pending = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._hooks = []
        self._last_launch_event = None         # HIP event at the launch of the step's last bucket (while the StallClock runs)
        self._tail_pairs = []                  # (that event, event behind finish()'s waits) per step
        self.enabled = ((self.world > 1 or (force and dist.is_initialized())) and self.comm is None) if self._abi is None \
            else self._abi.active
        # gloo with DEVICE tensors (only the one-GPU rehearsal of the multi-rank path uses it): ProcessGroupGloo stages every
        # collective through a private HIGH-PRIORITY stream that waits ON THE DEVICE for the compute stream.  With several
        # ranks sharing one GPU those waiting high-priority queues starve the normal-priority compute queues they wait for
        # (stuck-rank dumps in profiles/r3_dp4_hang/, tools/gloo_async_probe.py).  There the host drains
        # the compute stream before it hands a bucket to gloo, so no queue ever waits on the device; the collectives stay
        # asynchronous, issued from the hooks and waited in finish() exactly as under RCCL (whose stream has normal priority
        # and whose ranks own their GPU).
        self.clock = getattr(comm, "clock", None)       # StallClock of the trainer's communicator: exposed waits on the buckets
        self._drain_first = (not _ASYNC and self.enabled and self._abi is None and arena.flat_grad.is_cuda
                             and dist.get_backend(process_group) == "gloo")
        self._blocking = _BLOCKING
        if self.enabled:
            for p in arena.params:
                if p.requires_grad:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _add_bucket(self, start, end, members):
        b = len(self.buckets)
        self.buckets.append((start, end, len(members)))
        self.bucket_members.append([self._index_of[id(p)] for p in members])
        for p in members:
            self.param_bucket[id(p)] = b

    def start(self):
        """Call before backward."""
        self._pending = [n for (_, _, n) in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._handles = []
        self._last_launch_event = None

    def tail_ms(self) -> float:
        """Mean time on the compute stream between the launch of a step's LAST bucket (its gradients had just become ready) and the
        end of finish()'s waits - what clip / Adam waited for behind the end of backward.  Measured while the communicator's
        StallClock is enabled; synchronises; resets."""
        pairs, self._tail_pairs = self._tail_pairs, []
        if not pairs:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)

    @property
    def last_bucket_bytes(self) -> int:
        """Size of the bucket whose gradients are ready last (the front of the arena)."""
        s, e, _ = self.buckets[-1]
        return 4 * (e - s)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        if _TRACE:
            import sys, threading
            print(f"[ddp rank {self._rank}] launch bucket {b} of {len(self.buckets)} [{s}:{e}) from "
                  f"{threading.current_thread().name}", file=sys.stderr, flush=True)
        self.arena.gather(self.bucket_members[b])      # gather mode: the bucket's gradients move into the arena now
        if self.clock is not None and self.clock.enabled and self.arena.flat_grad.is_cuda:
            ev = torch.cuda.Event(enable_timing=True)   # overwritten by every launch: what is left is the step's last one
            ev.record()
            self._last_launch_event = ev
        if self._abi is not None:
            self._handles.append(self._abi.all_reduce_async(self.arena.flat_grad[s:e]))
            return
        if self._drain_first:
            torch.cuda.current_stream().synchronize()
        if self._blocking:
            dist.all_reduce(self.arena.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.pg)
            return
        self._handles.append(dist.all_reduce(self.arena.flat_grad[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def _on_grad(self, p):
        if not self.enabled:
            return
        b = self.param_bucket[id(p)]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launch(b)

    def finish(self):
        """Call after backward: flush buckets that are still waiting (parameters without gradient),
        wait for every collective.  The arena then holds SUMS over ranks; the 1 / world factor is
        applied inside the clip / Adam kernel (``grad_scale``) unless ``average`` asks for it here."""
        if self.comm is not None:
            self.arena.gather()
            self.comm.all_reduce_sum(self.arena.flat_grad)
            if self.average:
                self.arena.flat_grad.mul_(1.0 / self.world)
            return
        if not self.enabled:
            self.arena.gather()
            return
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for i, h in enumerate(self._handles):
            if self.clock is not None:
                self.clock.bracket(h.wait)           # what the compute stream waits here was not hidden behind backward
            else:
                h.wait()
            if _TRACE:
                import sys
                print(f"[ddp rank {self._rank}] handle {i} of {len(self._handles)} done", file=sys.stderr, flush=True)
        self._handles = []
        if self._last_launch_event is not None:
            done = torch.cuda.Event(enable_timing=True)
            done.record()
            self._tail_pairs.append((self._last_launch_event, done))
            self._last_launch_event = None
        if self.average:
            self.arena.flat_grad.mul_(1.0 / self.world)

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


def broadcast_arena(arena: FlatArena, src: int = 0, process_group=None):
    """Make every replica start from rank ``src``'s weights."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(arena.flat_param, src=src, group=process_group)
