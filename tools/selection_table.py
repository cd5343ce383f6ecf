#!/usr/bin/env python3
"""Which kernel the library picks by lattice, slab count and halo transport: prints one line per point, in the form
tests/test_gpu_parity.py::test_kernel_selection_table pins (run on an MI355X; no lattice is advanced).
    python tools/selection_table.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import advanced_hpc_lbm_amd as L  # noqa: E402

EX = {"none": L.EXCHANGE_AUTO, "copy": L.EXCHANGE_COPY, "p2p": L.EXCHANGE_P2P, "rccl": L.EXCHANGE_RCCL}
POINTS = [
    # lone lattices
    (128, 128, 1, "none"), (128, 256, 1, "none"), (256, 256, 1, "none"), (1024, 1024, 1, "none"), (100, 100, 1, "none"),
    (1000, 600, 1, "none"), (2048, 2048, 1, "none"), (2050, 2048, 1, "none"), (4096, 4096, 1, "none"), (5120, 5120, 1, "none"),
    (8192, 8192, 1, "none"), (8192, 1024, 1, "none"), (48, 4096, 1, "none"), (64, 8, 1, "none"), (1024, 128, 1, "none"),
    # slabs of one process on one GPU
    (1024, 1024, 2, "copy"), (1024, 1024, 8, "copy"), (1024, 1024, 8, "p2p"), (8192, 8192, 2, "p2p"), (8192, 8192, 4, "p2p"),
    (8192, 8192, 8, "p2p"), (8192, 8192, 8, "copy"), (4096, 4096, 4, "p2p"), (2048, 2048, 2, "copy"), (1000, 600, 3, "copy"),
    (256, 256, 4, "p2p"), (6144, 6144, 1, "none"), (4096, 4096, 2, "p2p"),
    # one rank of a RCCL job, as a ring of one (LBM_FORCE_EXCHANGE): the slab this rank would hold at N = 8 / 4, and a narrow one
    (8192, 1024, 1, "rccl"), (8192, 2048, 1, "rccl"), (1024, 128, 1, "rccl"),
]


def probe(nx, ny, nslabs, ex):
    p = L.Param(nx, ny, 10, 10, 0.1, 0.01, 1.85)
    ob = np.zeros((ny, nx), dtype=np.int32)
    if ex == "rccl":
        os.environ["LBM_FORCE_EXCHANGE"] = "1"
        try:
            lat = L.Lattice(p, ob, rank=0, nranks=1, device=0, unique_id=L.rccl_unique_id(), exchange=L.EXCHANGE_RCCL)
        finally:
            del os.environ["LBM_FORCE_EXCHANGE"]
    else:
        lat = L.Lattice(p, ob, nslabs=nslabs, devices=[0] * nslabs, exchange=EX[ex])
    with lat:
        tb = int(lat.info("time_block_active"))
        wave = int(lat.info("march_kernel"))
        return (int(lat.info("engine_next")), tb, wave, int(lat.info("wave_cols_active")) if (wave and tb >= 4) else 0)


if __name__ == "__main__":
    for pt in POINTS:
        print(f"    ({pt[0]}, {pt[1]}, {pt[2]}, \"{pt[3]}\"): {probe(*pt)},", flush=True)
