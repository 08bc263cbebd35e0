"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Checks that
FlatArena + GradReducer (mqgan_amd/ddp.py) produce exactly the sum of the per-rank gradients, with
several buckets, hooks firing during backward, and a parameter that never gets a gradient
(`hidden_proj` in the real model) not stalling its bucket."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(13, 37)
        self.b = torch.nn.Linear(37, 41)
        self.unused = torch.nn.Linear(5, 3)          # never receives a gradient
        self.c = torch.nn.Linear(41, 2)
        self.scale = torch.nn.Parameter(torch.tensor(1.5))   # 0-d parameter, like APTx beta/gamma

    def forward(self, x):
        return self.c(torch.tanh(self.b(torch.relu(self.a(x))))) * self.scale


class AsyncBucketDouble:
    """A communicator with comm.RcclAbiComm's BUCKET interface (``buckets_async``, ``all_reduce_async`` -> handle with wait()),
    carried by gloo so that two CPU ranks can drive the branch of GradReducer that hands its buckets to mq_allreduce_bucket on
    the GPU (one forced rank there: tests/test_gpu_exchange.py)."""
    buckets_async = True

    def __init__(self):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.calls = 0

    @property
    def active(self):
        return self.world > 1

    def all_reduce_async(self, t):
        self.calls += 1
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def _worker(rank, world, port, out_q, gather=False, abi=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mqgan_amd.ddp import FlatArena, GradReducer, broadcast_arena
        torch.manual_seed(100 + rank)                # different init per rank: broadcast must fix that
        net = Net()
        arena = FlatArena([net], gather=gather)
        broadcast_arena(arena)
        double = AsyncBucketDouble() if abi else None
        red = GradReducer(arena, bucket_bytes=1024, tail_bytes=64, comm=double)  # several buckets, tapering towards the front of the arena
        assert len(red.buckets) >= 3
        assert red.enabled and (red._abi is double) and red.world == world and red.comm is None
        launched_during_backward = []
        orig = red._launch
        red._launch = lambda b: (launched_during_backward.append(b), orig(b))[1]
        results = []
        for step in range(2):
            arena.zero_grad()
            red.start()
            torch.manual_seed(7 + 10 * step + rank)
            x = torch.randn(6, 13)
            net(x).pow(2).sum().backward()
            n_hook = len(launched_during_backward)
            red.finish()
            results.append((arena.flat_grad.clone(), n_hook))
        if abi:                                      # every bucket of both steps went through the communicator's own all-reduce
            assert double.calls == 2 * len(red.buckets), (double.calls, len(red.buckets))
        # every parameter still aliases the arena and .grad views were filled in place (gather mode: the gradients stay in
        # autograd's own tensors and were copied into the arena bucket by bucket)
        for p, o in zip(arena.params, arena.offsets):
            assert p.data_ptr() == arena.flat_param.data_ptr() + 4 * o
            if not gather:
                assert p.grad.data_ptr() == arena.flat_grad.data_ptr() + 4 * o
        out_q.put((rank, arena.flat_param.clone(), [r[0] for r in results], [r[1] for r in results]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather,abi", [(False, False), (True, False), (True, True)],
                         ids=["torch.distributed", "torch.distributed-gather", "async-bucket communicator (the C ABI route)"])
def test_bucketed_allreduce_world2(gather, abi):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, gather, abi)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=60) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    (_, w0, g0, hooks0), (_, w1, g1, hooks1) = got
    assert torch.equal(w0, w1), "broadcast_arena must make the replicas identical"
    for a, b in zip(g0, g1):
        assert torch.equal(a, b), "all ranks must hold the same reduced gradient"
    assert min(hooks0 + hooks1) >= 1, "at least one bucket must be launched from a hook, i.e. during backward"

    # single-process reference: sum of the two ranks' gradients on the broadcast weights
    from mqgan_amd.ddp import FlatArena
    torch.manual_seed(100)
    net = Net()
    arena = FlatArena([net])
    assert torch.equal(arena.flat_param, w0)
    for step in range(2):
        total = torch.zeros_like(arena.flat_grad)
        for rank in range(world):
            arena.zero_grad()
            torch.manual_seed(7 + 10 * step + rank)
            net(torch.randn(6, 13)).pow(2).sum().backward()
            total += arena.flat_grad
        torch.testing.assert_close(g0[step], total, rtol=1e-6, atol=1e-6)
    # the unused layer's slice is all zeros (flushed, not stalled)
    i = [id(p) for p in arena.params].index(id(net.unused.weight))
    o = arena.offsets[i]
    assert float(g0[0][o:o + net.unused.weight.numel()].abs().max()) == 0.0


def test_buckets_taper_towards_the_last_ready_gradients():
    """Bucket geometry (pure host logic): the buckets tile the gradient arena without gaps in reverse registration order, none is
    larger than the cap plus one tensor, each closes at half of what is left in front of it, and the bucket at the FRONT of the
    arena - the first layers, whose gradients are ready when backward ends, so its all-reduce is never hidden - is the smallest."""
    from mqgan_amd.ddp import FlatArena, GradReducer
    net = torch.nn.Sequential(*[torch.nn.Linear(64, 64) for _ in range(40)])           # 40 x (4096 + 64) floats = 166,400
    arena = FlatArena([net])
    cap, tail = 16384 * 4, 1024 * 4                                                    # bytes
    red = GradReducer(arena, bucket_bytes=cap, tail_bytes=tail)
    sizes = [e - s for (s, e, _) in red.buckets]
    assert red.buckets[0][1] == arena.numel and red.buckets[-1][0] == 0
    for (s0, _, _), (_, e1, _) in zip(red.buckets, red.buckets[1:]):
        assert e1 == s0                                                                # contiguous, back to front
    biggest = 4096
    assert max(sizes) <= cap // 4 + biggest
    for (s, e, _), n in zip(red.buckets[:-1], sizes):
        want = min(cap // 4, max(tail // 4, e // 2))
        assert want <= n <= want + biggest, (s, e, n, want)
    assert sizes[-1] == min(sizes) and sizes[-1] <= tail // 4 + biggest, sizes
    assert red.last_bucket_bytes == 4 * sizes[-1]
    assert sum(len(m) for m in red.bucket_members) == len(arena.params)
    # r4's geometry for comparison: equal caps leave an arbitrary remainder to the end
    flat = GradReducer(arena, bucket_bytes=cap, tail_bytes=cap)
    assert len(red.buckets) > len(flat.buckets)


def test_arena_survives_load_state_dict_and_zero_grad():
    from mqgan_amd.ddp import FlatArena
    net = Net()
    arena = FlatArena([net])
    sd = {k: v.clone() + 1 for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    arena.rebind()
    for p, o in zip(arena.params, arena.offsets):
        assert p.data_ptr() == arena.flat_param.data_ptr() + 4 * o
    assert torch.equal(net.a.weight, sd["a.weight"])
    net.zero_grad(set_to_none=True)           # something resets .grad: zero_grad() re-attaches the views
    arena.zero_grad()
    net(torch.randn(2, 13)).sum().backward()
    assert float(arena.flat_grad.abs().sum()) > 0


def _norm_worker(rank, world, port, out_q):
    """The global-normaliser protocol of the data-parallel step on two ragged shards (plain torch on CPU, gloo):
    per-rank numerators and counts -> TorchComm.all_reduce_sum -> loss from the reduced sums -> local backward with the
    GLOBAL count -> SUM all-reduce of the gradients.  Must equal the single-process gradient on the whole batch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mqgan_amd.comm import TorchComm, default_comm
        from mqgan_amd.ddp import FlatArena, GradReducer
        comm = default_comm()
        assert isinstance(comm, TorchComm) and comm.world == world and comm.rank == rank
        torch.manual_seed(3)
        net = Net()
        arena = FlatArena([net])
        red = GradReducer(arena, bucket_bytes=1024)
        torch.manual_seed(11)
        x_all, valid_all = torch.randn(7, 13), (torch.rand(7, 2) > 0.35).float()
        lo, hi = (0, 3) if rank == 0 else (3, 7)              # unequal shards: 3 and 4 rows, different valid counts
        x, valid = x_all[lo:hi], valid_all[lo:hi]
        arena.zero_grad()
        red.start()
        y = net(x)
        sums = torch.stack([((y - 1.0) ** 2 * valid).sum().detach(), valid.sum()])
        comm.all_reduce_sum(sums)                             # [global numerator, global count]
        loss_value = sums[0] / sums[1]                        # every rank holds the GLOBAL loss
        local = ((y - 1.0) ** 2 * valid).sum() / sums[1]      # its own numerator over the global count
        local.backward()
        red.finish()                                          # SUM over ranks, no 1 / world factor
        flag = comm.any_rank(rank == 1)                       # "my batch failed to load" on one rank only ...
        assert comm.any_rank(False) is False
        out_q.put((rank, float(loss_value), arena.flat_grad.clone(), 1.0 if flag else 0.0))
    finally:
        dist.destroy_process_group()


def test_global_normalisers_protocol_world2():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_norm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=60) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from mqgan_amd.ddp import FlatArena
    torch.manual_seed(3)
    net = Net()
    arena = FlatArena([net])
    torch.manual_seed(11)
    x_all, valid_all = torch.randn(7, 13), (torch.rand(7, 2) > 0.35).float()
    arena.zero_grad()
    loss = ((net(x_all) - 1.0) ** 2 * valid_all).sum() / valid_all.sum()
    loss.backward()
    for rank, lv, grad, flag in got:
        assert abs(lv - loss.item()) < 1e-6 * max(1.0, abs(loss.item()))
        torch.testing.assert_close(grad, arena.flat_grad, rtol=1e-5, atol=1e-6)
        assert flag == 1.0                                    # ... is seen by every rank: all of them skip the step
    assert torch.equal(got[0][2], got[1][2])
