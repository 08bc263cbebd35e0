"""The one collective the data-parallel step needs besides the gradient buckets: a SUM all-reduce of small device
buffers (loss numerators / counts).  ``Comm`` is the single-process identity; ``TorchComm`` runs on
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU); ``RcclAbiComm`` runs the device collectives through
the C ABI's own RCCL entry points (include/mqgan_hip.h: mq_comm_* / mq_allreduce_bucket), the path a non-Python consumer of the
library takes.  Tests may supply their own."""
import ctypes as C

import torch
import torch.distributed as dist


class Comm:
    world = 1
    rank = 0

    @property
    def active(self) -> bool:
        """Whether all_reduce_sum has to be called at all (callers skip the identity)."""
        return self.world > 1

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks; returns ``t``."""
        return t

    def all_reduce_sum_host(self, t: torch.Tensor) -> torch.Tensor:
        """The same for a HOST tensor, without touching the device stream (control decisions: no stream synchronisation)."""
        return self.all_reduce_sum(t)

    def any_rank(self, flag: bool) -> bool:
        """True on every rank if ``flag`` is true on any.  Host-side agreement: the device stream is not synchronised, so the
        host keeps running ahead of the GPU (the per-step skip decision of the training loop)."""
        if not self.active:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0])
        self.all_reduce_sum_host(t)
        return bool(t[0] > 0)


class StallClock:
    """How long the compute stream STALLS on collectives: a pair of HIP events around every point where the stream has to
    wait for one (the blocking loss-sum all-reduces; the waits on the asynchronous gradient buckets).  Nothing else is enqueued
    between the two events of a pair, so their distance is communication the step could not hide behind compute - the
    "exposed" part.  Off by default (``enabled``); ``read_ms()`` synchronises and returns the sum since the last read."""

    def __init__(self):
        self.enabled = False
        self._pairs = []

    def bracket(self, fn):
        if not (self.enabled and torch.cuda.is_available()):
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self._pairs.append((a, b))
        return out

    def read_ms(self) -> float:
        if not self._pairs:
            return 0.0
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._pairs)
        self._pairs = []
        return total


class TorchComm(Comm):
    """``force``: issue the collectives even in a group of one rank (a one-GPU box can then execute the whole RCCL path:
    communicator set-up, the process group's side stream, its ordering against the compute stream).

    Construct it on EVERY rank of ``group`` (default: the world) at the same point of the program: under RCCL it creates the
    gloo twin that carries host-side decisions, and creating a process group is itself a collective."""

    def __init__(self, group=None, force: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("TorchComm needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.force = bool(force)
        self.calls = 0
        self.clock = StallClock()
        # RCCL moves device memory only: host decisions (Comm.any_rank) travel over a gloo twin of the group.  Created HERE,
        # where every rank of the group is known to be (new_group is a collective over the WHOLE default group, so ranks
        # outside ``group`` must call it as well - see the class docstring): created lazily inside the training loop, a rank
        # whose loader ran dry would never reach the call while its peers block in it.
        self._host_group = None
        if self.active and dist.get_backend(self.group) != "gloo":
            ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
            self._host_group = dist.new_group(ranks=ranks, backend="gloo")

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def all_reduce_sum(self, t):
        if self.active:
            self.clock.bracket(lambda: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group))
            self.calls += 1
        return t

    def all_reduce_sum_host(self, t):
        if not self.active:
            return t
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._host_group if self._host_group is not None else self.group)
        return t


class _AbiHandle:
    """An all-reduce in flight on the communicator's side stream; wait() makes the CURRENT stream wait for it (no host block)."""

    def __init__(self, done):
        self.done = done

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)


class RcclAbiComm(Comm):
    """Device collectives through the library's C ABI (mq_comm_init / mq_allreduce_bucket; RCCL resolved by the library at run
    time), one communicator per process = per GPU.  ``world`` > 1 needs an initialised torch.distributed group of any backend for
    the HOST side only: the 128-byte RCCL id travels from rank 0 by ``broadcast_object_list`` and the per-step host decisions
    (``any_rank``) over a gloo group, exactly as under ``TorchComm``.  With ``world`` == 1 nothing but the library is involved;
    ``force`` then issues every collective anyway (a one-GPU box executes communicator set-up, the side stream and its event
    ordering end to end: tests/test_gpu_rccl.py).

    ``buckets_async``: ddp.GradReducer hands its buckets to ``all_reduce_async`` from the gradient hooks (side stream, ordered
    behind the bucket's last gradient by an event) and waits for the handles before clip / Adam."""

    buckets_async = True

    def __init__(self, world: int = None, rank: int = None, force: bool = False, group=None):
        from . import _lib
        self._lib = _lib
        self.lib = _lib.load()
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world, self.rank, self.force, self.group = int(world), int(rank), bool(force), group
        self.calls = 0
        self.clock = StallClock()
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(self.lib.mq_comm_unique_id(ident), "mq_comm_unique_id")
        self._host_group = None
        if self.world > 1:
            if not dist.is_initialized():
                raise RuntimeError("RcclAbiComm with world > 1 needs torch.distributed for the host-side exchange of the RCCL id")
            box = [ident.raw if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = C.create_string_buffer(box[0], 128)
            if dist.get_backend(group) != "gloo":
                ranks = dist.get_process_group_ranks(group) if group is not None else None
                self._host_group = dist.new_group(ranks=ranks, backend="gloo")
        self._comm = C.c_void_p()
        _lib.check(self.lib.mq_comm_init(C.byref(self._comm), self.world, self.rank, ident), "mq_comm_init")
        self._side = torch.cuda.Stream()

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def all_reduce_sum(self, t):
        """Blocking in stream order: on the current stream."""
        if self.active:
            self.clock.bracket(lambda: self._lib.check(
                self.lib.mq_allreduce_bucket(self._comm, self._lib.ptr(t, torch.float32), t.numel(), self._lib.stream()),
                "mq_allreduce_bucket"))
            self.calls += 1
        return t

    def all_reduce_async(self, t) -> _AbiHandle:
        """SUM of ``t`` over the ranks on the side stream, behind everything the current stream has been given so far."""
        self._side.wait_event(torch.cuda.current_stream().record_event())
        self._lib.check(self.lib.mq_allreduce_bucket(self._comm, self._lib.ptr(t, torch.float32), t.numel(),
                                                     C.c_void_p(self._side.cuda_stream)), "mq_allreduce_bucket")
        self.calls += 1
        return _AbiHandle(self._side.record_event())

    def all_reduce_sum_host(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._host_group if self._host_group is not None else self.group)
        return t

    def close(self):
        if self._comm:
            torch.cuda.synchronize()
            self._lib.check(self.lib.mq_comm_destroy(self._comm), "mq_comm_destroy")
            self._comm = C.c_void_p()


def default_comm() -> Comm:
    return TorchComm() if dist.is_initialized() and dist.get_world_size() > 1 else Comm()
