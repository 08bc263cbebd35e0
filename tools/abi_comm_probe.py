"""What one mq_allreduce_bucket call costs the HOST and the GPU with one forced rank: the same communicator driven on one side
stream only, and alternating between the side stream and the compute stream (as a training step does: gradient buckets on the side
stream, loss sums on the compute stream).  Usage: python tools/abi_comm_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

ge.build()
from mqgan_amd import _lib  # noqa: E402

torch.cuda.set_device(0)
lib = _lib.load()
ident = C.create_string_buffer(128)
_lib.check(lib.mq_comm_unique_id(ident))
comm = C.c_void_p()
t0 = time.perf_counter()
_lib.check(lib.mq_comm_init(C.byref(comm), 1, 0, ident))
print(f"mq_comm_init: {time.perf_counter() - t0:.2f} s")
side = torch.cuda.Stream()
cur = torch.cuda.current_stream()


def run(sizes, streams, label):
    bufs = [torch.randn(n, device="cuda") for n in sizes]
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for b, s in zip(bufs, streams):
        t = time.perf_counter()
        _lib.check(lib.mq_allreduce_bucket(comm, C.c_void_p(b.data_ptr()), b.numel(), C.c_void_p(s.cuda_stream)))
        host.append((time.perf_counter() - t) * 1e3)
    t_issue = (time.perf_counter() - t_all) * 1e3
    torch.cuda.synchronize()
    t_done = (time.perf_counter() - t_all) * 1e3
    print(f"{label}: {len(sizes)} calls, host per call min {min(host):.3f} / median {sorted(host)[len(host) // 2]:.3f} / max {max(host):.3f} ms; "
          f"issued in {t_issue:.2f} ms, complete after {t_done:.2f} ms")


MiB = 1 << 18      # floats
sizes = [35 * MiB, 37 * MiB, 16 * MiB, 6 * MiB, 1024, 39 * MiB, 34 * MiB, 24 * MiB, 11 * MiB, 6 * MiB, 3 * MiB, 50000]
for rep in range(3):
    run(sizes, [side] * len(sizes), f"side stream only (pass {rep})")
for rep in range(3):
    run(sizes, [side if i % 2 == 0 else cur for i in range(len(sizes))], f"alternating streams (pass {rep})")
run([16] * 12, [cur] * 12, "16-float sums on the compute stream")
_lib.check(lib.mq_comm_destroy(comm))
