"""The one collective the data-parallel step needs besides the gradient buckets: a SUM all-reduce of small device
buffers (loss numerators / counts).  ``Comm`` is the single-process identity; ``TorchComm`` runs on
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU).  Tests may supply their own."""
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


def default_comm() -> Comm:
    return TorchComm() if dist.is_initialized() and dist.get_world_size() > 1 else Comm()
