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


class TorchComm(Comm):
    """``force``: issue the collectives even in a group of one rank (a one-GPU box can then execute the whole RCCL path:
    communicator set-up, the process group's side stream, its ordering against the compute stream)."""

    def __init__(self, group=None, force: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("TorchComm needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.force = bool(force)
        self.calls = 0
        self._host_group = None

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def all_reduce_sum(self, t):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.calls += 1
        return t


    def all_reduce_sum_host(self, t):
        if not self.active:
            return t
        if dist.get_backend(self.group) == "gloo":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            return t
        if self._host_group is None:               # RCCL moves device memory only: host decisions travel over a gloo twin
            ranks = dist.get_process_group_ranks(self.group) if self.group is not None else None
            self._host_group = dist.new_group(ranks=ranks, backend="gloo")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._host_group)
        return t


def default_comm() -> Comm:
    return TorchComm() if dist.is_initialized() and dist.get_world_size() > 1 else Comm()
