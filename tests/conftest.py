import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def variants_lib():
    """libmqgan_hip_variants.so (make -C mqgan_amd/csrc variants): the product code plus the transform building blocks as exported
    entry points, the measured-and-rejected kernel variants and the fused-Winograd prototype - what some bit-identity tests use as
    their reference.  Built on first use (a no-op when the objects travelled with the tree)."""
    import __graft_entry__ as g
    g.build()
    g.build_variants()
    from mqgan_amd import _lib
    return _lib.load_variants()
