"""-m gpu: data-parallel EXACTNESS on one GPU.  Two shards of a ragged batch go through two Trainer
instances that stand for two ranks; their only link is a world-2 communicator (the same `Comm` interface the real run
implements over torch.distributed / RCCL).  The two "ranks" are threads of this process that run strictly one at a
time and hand over at every all-reduce, so the whole exchange - loss numerators and counts, the LeCam EMA chain, the
gradient sum - happens exactly as it would between processes.  After each optimisation step every rank must hold

    * the same losses as ONE trainer stepping on the concatenated (global) batch,
    * a gradient arena equal to the global-batch gradient (1e-5 relative), i.e. sum over ranks == global,
    * the same LeCam EMA state and the same updated weights.
"""
import threading

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

from gpu_util import DEV, npz_state_dict, rel_err  # noqa: E402
from test_oracle_trainstep import TINY_CFG  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()
    assert torch.cuda.is_available()


class _Shared:
    def __init__(self, world):
        self.world = world
        self.run = threading.Lock()            # the baton: whoever holds it executes model code
        self.cv = threading.Condition()
        self.slots = [None] * world
        self.arrived = 0
        self.gen = 0
        self.calls = 0


def make_comm(shared, rank):
    from mqgan_amd.comm import Comm

    class BatonComm(Comm):
        world = shared.world

        def __init__(self):
            self.rank = rank

        def all_reduce_sum(self, t):
            shared.run.release()
            try:
                with shared.cv:
                    shared.slots[rank] = t
                    shared.arrived += 1
                    if shared.arrived == shared.world:
                        total = shared.slots[0].clone()
                        for o in shared.slots[1:]:
                            assert o.shape == total.shape, "mismatched collectives across ranks"
                            total += o
                        for o in shared.slots:
                            o.copy_(total)
                        shared.arrived = 0
                        shared.gen += 1
                        shared.calls += 1
                        shared.cv.notify_all()
                    else:
                        g = shared.gen
                        while shared.gen == g:
                            if not shared.cv.wait(timeout=120):
                                raise RuntimeError("rank %d: peer never reached the all-reduce" % rank)
            finally:
                shared.run.acquire()
            return t

    return BatonComm()


def _trainer(g, use_fm, comm=None):
    from mqgan_amd.train import Trainer
    cfg = {"model": TINY_CFG["model"], "training": dict(TINY_CFG["training"], use_fm_loss=use_fm, pretrained=None, precision="fp32")}
    tr = Trainer(cfg, device=DEV, build_data=False, comm=comm)
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


@pytest.mark.parametrize("use_fm", [False, True])
def test_two_ragged_shards_equal_the_global_batch(use_fm):
    from mqgan_amd.train import synthetic_mel
    g = load_golden("g7_trainstep.npz")
    T, mel = 40, 16
    xs = torch.from_numpy(np.stack([synthetic_mel(T, mel, 300 + b) for b in range(5)])).to(DEV)
    lens_all = torch.tensor([40, 23, 31, 40, 17], dtype=torch.int32, device=DEV)
    for b in range(5):
        xs[b, int(lens_all[b]):] = 0.0
    shards = [(xs[:2].contiguous(), lens_all[:2].contiguous()), (xs[2:].contiguous(), lens_all[2:].contiguous())]   # 2 + 3 items: unequal shards

    # ---- one trainer on the global batch
    ref = _trainer(g, use_fm)
    ref_out = []
    for _ in range(2):                        # step 0: train-mode discriminators; step 1: eval mode, real + fake as one batch
        out = ref.train_step(xs, lens_all)
        ref_out.append(({k: float(v) for k, v in out.items()}, ref.arena_g.flat_grad.clone(), ref.arena_d.flat_grad.clone(),
                        ref.gan_loss._state.clone(), ref.arena_g.flat_param.clone(), ref.arena_d.flat_param.clone()))

    # ---- two ranks
    shared = _Shared(2)
    got, errors = [None, None], []

    def rank_main(r):
        with shared.run:
            try:
                tr = _trainer(g, use_fm, comm=make_comm(shared, r))
                assert tr.world == 2 and tr.rank == r
                res = []
                for _ in range(2):
                    out = tr.train_step(*shards[r])
                    res.append(({k: float(v) for k, v in out.items()}, tr.arena_g.flat_grad.clone(), tr.arena_d.flat_grad.clone(),
                                tr.gan_loss._state.clone(), tr.arena_g.flat_param.clone(), tr.arena_d.flat_param.clone()))
                got[r] = res
            except BaseException as e:        # noqa: BLE001 - reported by the main thread
                errors.append((r, e))
                with shared.cv:               # let the peer fail fast instead of waiting for us
                    shared.gen += 1
                    shared.cv.notify_all()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join(timeout=300)
    assert not errors, errors
    assert got[0] is not None and got[1] is not None
    assert shared.calls >= 2 * (2 + 1 + 2)     # per step: 2 (D sums, LeCam) + 1+ (G losses) + 2 gradient arenas

    for s in range(2):
        want = ref_out[s]
        for r in range(2):
            have = got[r][s]
            for k in ("loss_d", "loss_g_total", "loss_recon_pre", "loss_recon_post", "loss_gan", "loss_fm"):
                np.testing.assert_allclose(have[0][k], want[0][k], rtol=2e-5, atol=1e-7, err_msg=f"step {s} rank {r} {k}")
            assert rel_err(have[1], want[1]) < 2e-5, f"step {s} rank {r}: generator gradient {rel_err(have[1], want[1]):.2e}"
            assert rel_err(have[2], want[2]) < 2e-5, f"step {s} rank {r}: discriminator gradient {rel_err(have[2], want[2]):.2e}"
            np.testing.assert_allclose(have[3].cpu().numpy(), want[3].cpu().numpy(), rtol=1e-5, atol=1e-7, err_msg="LeCam EMA state")
            assert rel_err(have[4], want[4]) < 1e-6 and rel_err(have[5], want[5]) < 1e-6, "updated weights"
        # replicas stay bit-identical to each other
        assert torch.equal(got[0][s][3], got[1][s][3])
        assert torch.equal(got[0][s][1], got[1][s][1]) and torch.equal(got[0][s][2], got[1][s][2])


# ----------------------------------------------------------------------------------------- the torch.distributed path
def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _dist_rank(rank, world, port, use_fm, q):
    """One rank of a real process group (gloo here; the production run differs only in backend = RCCL): TorchComm for the loss
    sums, GradReducer's post-accumulate hooks + per-bucket arena fill + asynchronous bucket all-reduces for the gradients."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as ge
        ge.build()
        from mqgan_amd.train import synthetic_mel
        from mqgan_amd import ddp
        g = load_golden("g7_trainstep.npz")
        T, mel = 40, 16
        xs = torch.from_numpy(np.stack([synthetic_mel(T, mel, 300 + b) for b in range(5)])).to(DEV)
        lens_all = torch.tensor([40, 23, 31, 40, 17], dtype=torch.int32, device=DEV)
        for b in range(5):
            xs[b, int(lens_all[b]):] = 0.0
        sl = slice(0, 2) if rank == 0 else slice(2, 5)
        tr = _trainer(g, use_fm)
        assert tr.world == 2 and tr.rank == rank and tr.reducer_g.enabled and tr.arena_g.gather_mode
        tr.reducer_g = ddp.GradReducer(tr.arena_g, bucket_bytes=64 << 10)       # several buckets on the tiny model
        tr.reducer_d = ddp.GradReducer(tr.arena_d, bucket_bytes=64 << 10)
        launched = []
        for red in (tr.reducer_g, tr.reducer_d):
            orig = red._launch
            red._launch = (lambda o: (lambda b: (launched.append(b), o(b))[1]))(orig)
        res = []
        for _ in range(2):
            out = tr.train_step(xs[sl].contiguous(), lens_all[sl].contiguous())
            res.append(({k: float(v) for k, v in out.items()}, tr.arena_g.flat_grad.cpu().numpy(), tr.arena_d.flat_grad.cpu().numpy(),
                        tr.gan_loss._state.cpu().numpy(), tr.arena_g.flat_param.cpu().numpy(), tr.arena_d.flat_param.cpu().numpy()))
        q.put((rank, res, len(launched), len(tr.reducer_g.buckets) + len(tr.reducer_d.buckets)))     # numpy: pickled by value
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_fm", [True])
def test_two_processes_over_torch_distributed_equal_the_global_batch(use_fm):
    import torch.multiprocessing as mp
    from mqgan_amd.train import synthetic_mel
    g = load_golden("g7_trainstep.npz")
    T, mel = 40, 16
    xs = torch.from_numpy(np.stack([synthetic_mel(T, mel, 300 + b) for b in range(5)])).to(DEV)
    lens_all = torch.tensor([40, 23, 31, 40, 17], dtype=torch.int32, device=DEV)
    for b in range(5):
        xs[b, int(lens_all[b]):] = 0.0
    ref = _trainer(g, use_fm)
    ref_out = []
    for _ in range(2):
        out = ref.train_step(xs, lens_all)
        ref_out.append(({k: float(v) for k, v in out.items()}, ref.arena_g.flat_grad.cpu(), ref.arena_d.flat_grad.cpu(),
                        ref.gan_loss._state.cpu(), ref.arena_g.flat_param.cpu(), ref.arena_d.flat_param.cpu()))
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_rank, args=(r, 2, port, use_fm, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = [(r, [(st[0],) + tuple(torch.from_numpy(a) for a in st[1:]) for st in res], nl, nb) for r, res, nl, nb in got]
    for rank, res, n_launched, n_buckets in got:
        assert n_buckets >= 4 and n_launched == 2 * n_buckets          # every bucket, every step
        for s in range(2):
            want, have = ref_out[s], res[s]
            for k in ("loss_d", "loss_g_total", "loss_recon_pre", "loss_recon_post", "loss_gan", "loss_fm"):
                np.testing.assert_allclose(have[0][k], want[0][k], rtol=2e-5, atol=1e-7, err_msg=f"step {s} rank {rank} {k}")
            assert rel_err(have[1], want[1]) < 2e-5 and rel_err(have[2], want[2]) < 2e-5, f"step {s} rank {rank}: gradients"
            np.testing.assert_allclose(have[3].numpy(), want[3].numpy(), rtol=1e-5, atol=1e-7, err_msg="LeCam EMA state")
            assert rel_err(have[4], want[4]) < 1e-6 and rel_err(have[5], want[5]) < 1e-6, "updated weights"
    for s in range(2):                                                  # replicas bit-identical to each other
        for i in (1, 2, 3, 4, 5):
            assert torch.equal(got[0][1][s][i], got[1][1][s][i])


# ----------------------------------------------------------------------------------------- the epoch loop under two ranks
def _epoch_rank(rank, world, port, data_dir, q):
    """Trainer._train_epoch under two real ranks (gloo, one GPU): DistributedSampler shards, per-rank random crop lengths
    (different T on the two ranks in the same step), PrefetchLoader, the host-side skip agreement (Comm.any_rank) when ONE
    rank's whole batch fails to load, gradient buckets + loss sums every step that is taken."""
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import __graft_entry__ as ge
        ge.build()
        from mqgan_amd.train import Trainer
        cfg = {"data": {"data_dir": data_dir, "output_dir": None, "validation_split": 0.0, "crop_len": [32, 48], "batch_size": 1,
                        "num_workers": 0},
               "model": TINY_CFG["model"], "training": dict(TINY_CFG["training"], num_epochs=1, pretrained=None),
               "logging": {"eval_interval": 1, "save_interval": 1, "log_interval": 1, "num_plot_examples": 1}}
        logs = []
        tr = Trainer(cfg, device=DEV, log_fn=logs.append)
        assert tr.world == 2 and tr.comm.active
        n_batches = len(tr.train_loader)
        tr._train_epoch(1)
        torch.cuda.synchronize()
        q.put((rank, n_batches, tr.global_step, tr.arena_g.flat_param.cpu().numpy(), tr.arena_d.flat_param.cpu().numpy(),
               tr.gan_loss._state.cpu().numpy(), [np.isfinite(r["loss_g_total"]) for r in logs]))
    except BaseException as e:                 # noqa: BLE001
        import traceback
        q.put((rank, "error", repr(e), traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_epoch_loop_two_ranks_skip_agreement(tmp_path):
    import torch.multiprocessing as mp
    from mqgan_amd.train import synthetic_mel
    data = tmp_path / "data"
    data.mkdir()
    for i, L in enumerate([64, 40, 96, 33, 70, 128, 50]):
        np.save(data / f"u{i}.npy", synthetic_mel(L, 16, i))
    np.save(data / "bad_rank1.npy", np.zeros(9, np.float32))              # not 2-D: dropped by the collate -> an EMPTY batch of 1
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_epoch_rank, args=(r, 2, port, str(data), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for g in got:
        assert g[1] != "error", g[2:]
    (_, nb0, steps0, g0, d0, s0, fin0), (_, nb1, steps1, g1, d1, s1, fin1) = got
    assert nb0 == nb1 == 4                                                # 8 files over two ranks, batch 1
    assert steps0 == steps1 == 3                                          # the step in which ONE rank drew the bad file was skipped by both
    assert all(fin0) and all(fin1)
    assert np.array_equal(g0, g1) and np.array_equal(d0, d1) and np.array_equal(s0, s1)   # replicas stay bit-identical
