"""-m gpu: the gradient exchange as it runs in production, as far as ONE GPU can execute it.

  * the RCCL path itself (backend "nccl", a group of one rank, every collective force-issued): post-accumulate hooks ->
    per-bucket mq_multi_gather_args -> asynchronous all-reduce on the process group's side stream -> wait -> clip / Adam,
    plus the loss-sum all-reduces - bit-identical to the step without any collective;
  * the gradient-arena fill with the host running far ahead of the stream (the r2 race: a pinned staging ring the host could
    wrap before the device had read it; now the rows travel by value in the kernel arguments);
  * twelve unsynchronised optimisation steps with a different crop length per step (reference train.py:140-197: the
    allocator hands out different gradient addresses every step) against the same steps synchronised one by one.
"""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

from gpu_util import DEV, npz_state_dict  # noqa: E402
from test_oracle_trainstep import TINY_CFG  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()


def _trainer(g, **kw):
    from mqgan_amd.train import Trainer
    cfg = {"model": TINY_CFG["model"], "training": dict(TINY_CFG["training"], use_fm_loss=True, pretrained=None, precision="fp32")}
    tr = Trainer(cfg, device=DEV, build_data=False, **kw)
    tr.generator.load_state_dict(npz_state_dict(g, "g0/"), strict=True)
    tr.patch_discriminator.load_state_dict(npz_state_dict(g, "p0/"), strict=True)
    tr.multibin_discriminator.load_state_dict(npz_state_dict(g, "m0/"), strict=True)
    tr.arena_g.rebind()
    tr.arena_d.rebind()
    tr.generator.eval()                       # no dropout: the step is a deterministic function of the batch
    tr.patch_discriminator.train()
    tr.multibin_discriminator.train()
    tr.epoch = 1
    return tr


def _batch(T, first, B=3, mel=16):
    from mqgan_amd.train import synthetic_mel
    x = torch.from_numpy(np.stack([synthetic_mel(T, mel, first + b) for b in range(B)])).to(DEV)
    lens = torch.tensor([T, max(8, T - 9), max(8, T // 2 + 1)][:B], dtype=torch.int32, device=DEV)
    for b in range(B):
        x[b, int(lens[b]):] = 0.0
    return x, lens


# ----------------------------------------------------------------------------------------- RCCL, one rank
def _rccl_rank(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import __graft_entry__ as ge
        ge.build()
        from mqgan_amd import ddp
        from mqgan_amd.comm import TorchComm
        g = load_golden("g7_trainstep.npz")
        ones = torch.ones(4, device=DEV)
        dist.all_reduce(ones)
        assert dist.get_backend() == "nccl" and float(ones.sum()) == 4.0
        c = TorchComm(force=True)                                    # host-side decisions travel over a gloo twin of the RCCL group
        assert c.any_rank(False) is False and c.any_rank(True) is True and c._host_group is not None
        Ts = [40, 24, 48, 32]
        batches = [_batch(T, 700 + 10 * i) for i, T in enumerate(Ts)]

        def run(force):
            tr = _trainer(g, force_collectives=force)
            launched = []
            if force:
                assert isinstance(tr.comm, TorchComm) and tr.comm.active and tr.world == 1
                tr.reducer_g.remove()
                tr.reducer_d.remove()
                tr.reducer_g = ddp.GradReducer(tr.arena_g, bucket_bytes=64 << 10, force=True)     # several buckets on the tiny model
                tr.reducer_d = ddp.GradReducer(tr.arena_d, bucket_bytes=64 << 10, force=True)
                for red in (tr.reducer_g, tr.reducer_d):
                    assert red.enabled and not red._blocking and len(red.buckets) >= 2
                    orig = red._launch
                    red._launch = (lambda o: (lambda b: (launched.append(b), o(b))[1]))(orig)
            else:
                assert not tr.reducer_g.enabled and not tr.comm.active
            res = []
            for x, lens in batches:             # no host synchronisation between the steps
                out = tr.train_step(x, lens)
                res.append((out, tr.arena_g.flat_grad.clone(), tr.arena_d.flat_grad.clone()))
            torch.cuda.synchronize()
            res = [({k: float(v) for k, v in o.items()}, a.cpu().numpy(), b.cpu().numpy()) for o, a, b in res]
            n_buckets = len(tr.reducer_g.buckets) + len(tr.reducer_d.buckets)
            calls = tr.comm.calls if force else 0
            return res, tr.arena_g.flat_param.cpu().numpy(), tr.arena_d.flat_param.cpu().numpy(), tr.gan_loss._state.cpu().numpy(), len(launched), n_buckets, calls

        plain = run(False)
        forced = run(True)
        q.put(("ok", plain, forced))
    except BaseException as e:                 # noqa: BLE001 - reported by the parent
        import traceback
        q.put(("error", repr(e), traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_rccl_single_rank_step_is_bit_identical_to_no_collectives():
    """Backend "nccl" (= RCCL on ROCm) with one rank and every collective forced: the hooks fire during backward, each bucket's
    gradients are moved by mq_multi_gather_args and all-reduced asynchronously on RCCL's stream, finish() waits, clip + Adam
    read the arena; the loss sums go through TorchComm.  A sum over one rank is the identity, so losses, both gradient arenas,
    LeCam state and weights after four unsynchronised steps (four crop lengths) must equal the plain single-process step BIT
    FOR BIT - any mis-ordering between the compute stream and RCCL's stream shows up as a difference."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_rank, args=(port, q))
    p.start()
    msg = q.get(timeout=420)
    p.join(timeout=120)
    assert msg[0] == "ok", msg[1:]
    assert p.exitcode == 0
    _, plain, forced = msg
    n_steps = len(plain[0])
    assert forced[4] == n_steps * forced[5] and forced[5] >= 4        # every bucket, every step, launched through _launch
    assert forced[6] >= n_steps * 3                                   # loss-sum all-reduces went through the process group
    for s in range(n_steps):
        assert plain[0][s][0] == forced[0][s][0], f"step {s}: losses differ"
        assert np.array_equal(plain[0][s][1], forced[0][s][1]), f"step {s}: generator gradient arena"
        assert np.array_equal(plain[0][s][2], forced[0][s][2]), f"step {s}: discriminator gradient arena"
    for i, what in ((1, "generator weights"), (2, "discriminator weights"), (3, "LeCam state")):
        assert np.array_equal(plain[i], forced[i]), what


# ----------------------------------------------------------------------------------------- RCCL behind the C ABI, one rank
def _abi_rank(q):
    try:
        import ctypes as C
        torch.cuda.set_device(0)
        import __graft_entry__ as ge
        ge.build()
        from mqgan_amd import _lib, ddp
        from mqgan_amd.comm import RcclAbiComm
        import torch.distributed as dist
        assert not dist.is_initialized()                               # nothing but the library's own entry points
        lib = _lib.load()
        # raw ctypes, as INTEGRATION.md section 2 binds it: id -> communicator -> in-place sum on a stream -> destroy
        ident = C.create_string_buffer(128)
        assert lib.mq_comm_unique_id(ident) == 0, lib.mq_last_error()
        assert any(ident.raw)
        comm = C.c_void_p()
        assert lib.mq_comm_init(C.byref(comm), 1, 0, ident) == 0, lib.mq_last_error()
        v = torch.randn(1 << 20, device=DEV)
        ref = v.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        assert lib.mq_allreduce_bucket(comm, C.c_void_p(v.data_ptr()), v.numel(), C.c_void_p(side.cuda_stream)) == 0, lib.mq_last_error()
        assert lib.mq_allreduce_bucket(comm, None, 0, None) == 0       # an empty bucket launches nothing
        torch.cuda.synchronize()
        assert torch.equal(v, ref)                                     # a sum over one rank
        assert lib.mq_allreduce_bucket(None, C.c_void_p(v.data_ptr()), 4, None) != 0 and b"comm is NULL" in lib.mq_last_error()
        assert lib.mq_comm_init(C.byref(comm), 2, 2, ident) != 0 and b"rank 2 of 2" in lib.mq_last_error()
        assert lib.mq_comm_destroy(comm) == 0

        g = load_golden("g7_trainstep.npz")
        Ts = [40, 24, 48, 32]
        batches = [_batch(T, 700 + 10 * i) for i, T in enumerate(Ts)]

        def run(abi):
            c = RcclAbiComm(force=True) if abi else None
            tr = _trainer(g, comm=c, force_collectives=abi)
            launched = []
            if abi:
                assert tr.comm is c and c.active and tr.world == 1 and c.any_rank(True) and not c.any_rank(False)
                tr.reducer_g.remove()
                tr.reducer_d.remove()
                tr.reducer_g = ddp.GradReducer(tr.arena_g, bucket_bytes=64 << 10, comm=c, force=True)
                tr.reducer_d = ddp.GradReducer(tr.arena_d, bucket_bytes=64 << 10, comm=c, force=True)
                for red in (tr.reducer_g, tr.reducer_d):
                    assert red.enabled and red._abi is c and len(red.buckets) >= 2
                    orig = red._launch
                    red._launch = (lambda o: (lambda b: (launched.append(b), o(b))[1]))(orig)
            else:
                assert not tr.reducer_g.enabled and not tr.comm.active
            res = []
            for x, lens in batches:             # no host synchronisation between the steps
                out = tr.train_step(x, lens)
                res.append((out, tr.arena_g.flat_grad.clone(), tr.arena_d.flat_grad.clone()))
            torch.cuda.synchronize()
            res = [({k: float(v) for k, v in o.items()}, a.cpu().numpy(), b.cpu().numpy()) for o, a, b in res]
            n_buckets = len(tr.reducer_g.buckets) + len(tr.reducer_d.buckets)
            calls = c.calls if abi else 0
            out = (res, tr.arena_g.flat_param.cpu().numpy(), tr.arena_d.flat_param.cpu().numpy(), tr.gan_loss._state.cpu().numpy(),
                   len(launched), n_buckets, calls)
            if abi:
                c.close()
            return out

        q.put(("ok", run(False), run(True)))
    except BaseException as e:                 # noqa: BLE001 - reported by the parent
        import traceback
        q.put(("error", repr(e), traceback.format_exc()))


def test_rccl_behind_the_c_abi_single_rank_step_is_bit_identical_to_no_collectives():
    """include/mqgan_hip.h: mq_comm_unique_id / mq_comm_init / mq_allreduce_bucket / mq_comm_destroy (SURVEY 8b `allreduce_bucket`),
    first through raw ctypes, then as the Trainer's communicator (comm.RcclAbiComm, NO torch.distributed in the process): the
    gradient hooks hand every bucket to mq_allreduce_bucket on the communicator's side stream behind an event, finish() makes the
    compute stream wait for the handles, the loss sums go through the same entry point on the compute stream.  One rank, every
    collective forced: four unsynchronised steps must equal the plain step bit for bit."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_abi_rank, args=(q,))
    p.start()
    msg = q.get(timeout=420)
    p.join(timeout=120)
    assert msg[0] == "ok", msg[1:]
    assert p.exitcode == 0
    _, plain, abi = msg
    n_steps = len(plain[0])
    assert abi[4] == n_steps * abi[5] and abi[5] >= 4                  # every bucket, every step
    assert abi[6] >= n_steps * (abi[5] + 3)                            # buckets + loss sums, all through mq_allreduce_bucket
    for s in range(n_steps):
        assert plain[0][s][0] == abi[0][s][0], f"step {s}: losses differ"
        assert np.array_equal(plain[0][s][1], abi[0][s][1]), f"step {s}: generator gradient arena"
        assert np.array_equal(plain[0][s][2], abi[0][s][2]), f"step {s}: discriminator gradient arena"
    for i, what in ((1, "generator weights"), (2, "discriminator weights"), (3, "LeCam state")):
        assert np.array_equal(plain[i], abi[i]), what


# ----------------------------------------------------------------------------------------- host far ahead of the stream
def test_arena_fill_rows_are_captured_at_launch():
    """Stall the stream, then enqueue 40 arena fills (each with freshly allocated gradient tensors, > 120 rows so that one fill
    is several launches) before the first one can have started.  Every fill must see ITS OWN table."""
    from mqgan_amd.ddp import FlatArena
    torch.manual_seed(5)

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(1 + (7 * i) % 300)) for i in range(260)])
    net = Net().to(DEV)
    arena = FlatArena([net], gather=True)
    rounds = 40
    keep, snaps = [], []
    grads = [[torch.randn_like(p) for p in net.ps] for _ in range(rounds)]       # distinct addresses per round
    torch.cuda.synchronize()
    torch.cuda._sleep(400_000_000)            # the stream is busy while the host enqueues everything below
    done = torch.cuda.Event()
    for r in range(rounds):
        arena.zero_grad()
        for p, gr in zip(net.ps, grads[r]):
            p.grad = gr
        arena.gather()
        snaps.append(arena.flat_grad.clone())
    done.record()
    ran_ahead = not done.query()
    torch.cuda.synchronize()
    for r in range(rounds):
        for p, o, gr in zip(arena.params, arena.offsets, grads[r]):
            assert torch.equal(snaps[r][o:o + p.numel()], gr.reshape(-1)), f"round {r}"
    assert ran_ahead, "the host did not get ahead of the stream: the stall was too short to test anything"


def test_unsynchronised_steps_with_varying_crop_length():
    """Twelve optimisation steps, a different T per step, issued back to back behind a stalled stream, against the same twelve
    steps with a synchronize() after each: final weights, moments and LeCam state identical bit for bit."""
    g = load_golden("g7_trainstep.npz")
    Ts = [40, 24, 48, 32, 56, 72, 40, 64, 24, 48, 32, 40]
    batches = [_batch(T, 900 + 10 * i) for i, T in enumerate(Ts)]

    def run(sync):
        tr = _trainer(g)
        torch.cuda.synchronize()
        if not sync:
            torch.cuda._sleep(400_000_000)
        for x, lens in batches:
            tr.train_step(x, lens)
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [t.clone() for t in (tr.arena_g.flat_param, tr.arena_d.flat_param, tr.optimizer_g.exp_avg, tr.optimizer_d.exp_avg_sq,
                                    tr.gan_loss._state)]

    a = run(True)
    b = run(False)
    for x, y, what in zip(a, b, ("generator weights", "discriminator weights", "generator exp_avg", "discriminator exp_avg_sq", "LeCam")):
        assert torch.equal(x, y), what


# ----------------------------------------------------------------------------------------- bench.py --gpus 2, end to end
def test_bench_self_launches_two_ranks_on_this_gpu():
    """`python bench.py --gpus 2` with no launcher in the environment: the parent starts two ranks, they rendezvous, count
    themselves with an all-reduce, build trainers, exchange gradient buckets and loss sums every step, and rank 0 prints ONE
    line with n_gpus = 2.  One GPU here, so the ranks share it over gloo (MQGAN_BENCH_SHARE_GPU=1, the rehearsal mode); the
    code path - launcher, hooks, per-bucket arena fill, asynchronous bucket all-reduces, TorchComm - is the N > 1 path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(MQGAN_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                          "--frames", "128", "--no-alt", "--no-configs", "--no-vq", "--no-cpu-baseline"], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout, out.stderr[-2000:])
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["losses_finite"] is True and line["config"]["global_batch"] == 8
    c = line["collectives"]
    assert c["backend"] == "gloo" and c["ranks_counted_by_all_reduce"] == 2 and c["asynchronous"] is True
    assert c["gradient_buckets_per_step"] >= 2 and c["loss_sum_all_reduces_total"] >= 3 * 3
    assert line["value"] > 0 and line["ms_per_step"] > 0
