#!/usr/bin/env python3
"""Root-cause probe for the multi-rank hang of round 2 (commit 3f05e3b): several ASYNCHRONOUS gloo all-reduces of DEVICE
tensors in flight with four ranks on one GPU.  Runs a ladder of variants, each as its own group of N processes with a
watchdog: a rank that makes no progress for WATCH seconds writes the Python stacks of all its threads and, for every OS
thread, its name / state / kernel wait channel (/proc/self/task/*) to OUT/<variant>_rank<r>.txt and exits with code 3.

    python tools/gloo_async_probe.py --ranks 4 --out gpurun_out/hang [--variants plain,thread,...]

Variants (all on cuda:0, backend gloo, 32 MiB buckets like GradReducer's):
  plain    main thread: 4 async all-reduces of slices of one flat device buffer, then wait() each; 6 rounds
  thread   the same collectives issued from a helper thread (as autograd's device thread does), waited in the main thread
  busy     plain + queued device work (8 f32 4096^3 matmuls) in front of every bucket: the collective's private HIGH-PRIORITY
           stream waits on the device for the compute stream
  busysync busy, but the host drains the compute stream before it issues each collective (no device-side wait)
  mixed    plain + a small blocking device all-reduce and a barrier between rounds (loss sums / bench barrier)
  cpu      plain on HOST tensors (control)
  block    plain with blocking calls (control)
"""
import argparse
import faulthandler
import os
import socket
import sys
import threading
import time


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _thread_table():
    rows = []
    base = "/proc/self/task"
    for tid in sorted(os.listdir(base), key=int):
        def rd(name):
            try:
                with open(f"{base}/{tid}/{name}") as f:
                    return f.read().strip()
            except OSError as e:
                return f"<{e.__class__.__name__}>"
        stat = rd("stat")
        state = stat.rsplit(")", 1)[-1].split()[0] if ")" in stat else "?"
        rows.append(f"tid {tid:>8}  {rd('comm'):<18} state {state}  wchan {rd('wchan')}  syscall {rd('syscall').split()[0] if rd('syscall') else ''}")
    return "\n".join(rows)


class Watch:
    def __init__(self, path, seconds):
        self.path, self.seconds, self.mark, self.t = path, seconds, "start", time.time()
        threading.Thread(target=self._run, daemon=True, name="probe-watchdog").start()

    def progress(self, mark):
        self.mark, self.t = mark, time.time()

    def _run(self):
        while True:
            time.sleep(1.0)
            if time.time() - self.t > self.seconds:
                with open(self.path, "w") as f:
                    f.write(f"NO PROGRESS for {self.seconds}s after: {self.mark}\n\n== OS threads ==\n{_thread_table()}\n\n== Python stacks ==\n")
                    f.flush()
                    faulthandler.dump_traceback(file=f, all_threads=True)
                os._exit(3)


def rank_main(rank, world, port, variant, out, watch_s):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    w = Watch(os.path.join(out, f"{variant}_rank{rank}.txt"), watch_s)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    on_dev = variant != "cpu"
    dev = torch.device("cuda", 0) if on_dev else torch.device("cpu")
    if on_dev:
        torch.cuda.set_device(0)
    n_b, per = 4, (32 << 20) // 4
    flat = torch.full((n_b * per,), float(rank + 1), device=dev)
    small = torch.ones(64, device=dev)
    a = torch.randn(4096, 4096, device=dev) if variant in ("busy", "busysync") else None
    want = float(sum(range(1, world + 1)))
    t_rounds = []
    for rnd in range(6):
        t_r = time.time()
        flat.fill_(float(rank + 1))
        handles = []

        def issue():
            for b in range(n_b):
                if a is not None:
                    for _ in range(8):
                        (a @ a)
                if variant == "busysync":
                    torch.cuda.current_stream().synchronize()
                sl = flat[b * per:(b + 1) * per]
                if variant == "block":
                    dist.all_reduce(sl)
                else:
                    handles.append(dist.all_reduce(sl, async_op=True))
                w.progress(f"round {rnd}: issued bucket {b}")
        if variant == "thread":
            t = threading.Thread(target=issue, name="probe-issuer")
            t.start()
            t.join()
        else:
            issue()
        for i, h in enumerate(handles):
            h.wait()
            w.progress(f"round {rnd}: waited handle {i}")
        if variant == "mixed":
            dist.all_reduce(small)
            dist.barrier()
            w.progress(f"round {rnd}: small all-reduce + barrier")
        if on_dev:
            torch.cuda.synchronize()
        got = (float(flat[0]), float(flat[-1]))
        assert got == (want, want), (rnd, got, want)
        w.progress(f"round {rnd} done")
        t_rounds.append(round(time.time() - t_r, 2))
    dist.barrier()
    dist.destroy_process_group()
    print(f"[probe {variant} rank {rank} of {world}] ok, seconds per round {t_rounds}", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=4)
    ap.add_argument("--out", default="gpurun_out/hang")
    ap.add_argument("--variants", default="cpu,block,plain,thread,busy,mixed")
    ap.add_argument("--watch", type=float, default=40.0)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    summary = {}
    for v in args.variants.split(","):
        port = _free_port()
        procs = [ctx.Process(target=rank_main, args=(r, args.ranks, port, v, args.out, args.watch)) for r in range(args.ranks)]
        t0 = time.time()
        for p in procs:
            p.start()
        codes = []
        for p in procs:
            p.join(timeout=max(1.0, args.watch * 3 - (time.time() - t0)))
            if p.exitcode is None:
                p.kill()                       # exactly the process started above
                p.join()
            codes.append(p.exitcode)
        summary[v] = {"exit_codes": codes, "seconds": round(time.time() - t0, 1), "hung": any(c != 0 for c in codes)}
        print(f"[probe] {v}: {summary[v]}", flush=True)
    import json
    with open(os.path.join(args.out, "summary.json"), "w") as f:
        json.dump({"ranks": args.ranks, "variants": summary}, f, indent=1)


if __name__ == "__main__":
    main()
