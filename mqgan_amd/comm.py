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

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    def all_reduce_sum(self, t):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            self.calls += 1
        return t


def default_comm() -> Comm:
    return TorchComm() if dist.is_initialized() and dist.get_world_size() > 1 else Comm()
