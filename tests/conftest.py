"""Shared fixtures.  `-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host
logic, ABI surface, gloo slab protocol.  `-m gpu` runs on an MI355X: parity of the HIP
path (through the C ABI) against the oracle, the golden files and lattice invariants."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "check")):
    if p not in sys.path:
        sys.path.insert(0, p)

DECKS = ("128x128", "128x256", "256x256", "1024x1024")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def deck_paths(deck):
    return os.path.join(ROOT, f"input_{deck}.params"), os.path.join(ROOT, f"obstacles_{deck}.dat")


@pytest.fixture(scope="session")
def oracle():
    import lbm_oracle as O
    return O.Oracle("strict")


@pytest.fixture(scope="session")
def O():
    import lbm_oracle
    return lbm_oracle


@pytest.fixture(scope="session")
def L():
    """The product package.  torch (if any test imported it) must be loaded first; see load_library()."""
    import advanced_hpc_lbm_amd as lbm
    lbm.load_library()
    return lbm


@pytest.fixture(scope="session")
def gpu(L):
    if L.device_count() < 1:
        pytest.fail("gpu-marked test started without a visible HIP device")
    return L


def load_kat(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


KATS = ("kat_8x6", "kat_16x12", "kat_33x20", "kat_64x40")
