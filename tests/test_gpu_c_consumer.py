"""-m gpu: examples/c_consumer - a plain-C99 program (gcc; HIP runtime C API for memory and streams) that calls libmqgan_hip.so through
include/mqgan_hip.h only: the codebook search (f32 kernel and screened search, bit for bit against the fmaf-chain definition restated
in C), one refiner-type convolution through mq_conv_plan / mq_conv_fprop_planned (the library picks Winograd) against a
double-precision loop, and the gradient-bucket all-reduce over RCCL with one rank on a side stream.  The drop-in boundary of SURVEY.md
8b exercised with no Python, PyTorch or C++ on the caller's side."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_consumer_runs_search_planned_conv_and_exchange():
    import __graft_entry__ as g
    g.build_native()
    ex = os.path.join(ROOT, "examples", "c_consumer")
    r = subprocess.run(["make", "-s", "-C", ex], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([os.path.join(ex, "c_consumer")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    out = r.stdout
    assert "c_consumer OK" in out
    assert "identical" in out and "MISMATCH" not in out
    assert 'form "winograd2d"' in out                       # the plan chose the refiner layers' form, not the direct kernel
    assert "unchanged (sum over one rank)" in out
