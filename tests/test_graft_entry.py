"""__graft_entry__: build() compiles everything (CPU-side check), smoke() runs on the GPU."""
import os

import pytest

from conftest import ROOT


def test_build_produces_all_artifacts():
    import __graft_entry__ as G
    G.build()
    assert os.path.exists(os.path.join(ROOT, "advanced-hpc-lbm_amd", "liblbm_mi355x.so"))
    assert os.path.exists(os.path.join(ROOT, "d2q9-bgk"))
    assert os.path.exists(os.path.join(ROOT, "oracle", "liblbm_oracle.so"))


@pytest.mark.gpu
def test_smoke(gpu):
    import __graft_entry__ as G
    G.smoke()
