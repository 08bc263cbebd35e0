"""The one collective the data-parallel step needs besides the gradient buckets: a SUM all-reduce of small device
buffers (loss numerators / counts).  ``Comm`` is the single-process identity; ``TorchComm`` runs on
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU).  Tests may supply their own."""
import torch
import torch.distributed as dist


class Comm:
    world = 1
    rank = 0

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks; returns ``t``."""
        return t


class TorchComm(Comm):
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("TorchComm needs an initialised torch.distributed process group")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce_sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def default_comm() -> Comm:
    return TorchComm() if dist.is_initialized() and dist.get_world_size() > 1 else Comm()
