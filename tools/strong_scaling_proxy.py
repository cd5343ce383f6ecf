#!/usr/bin/env python3
"""ONE rank's slab of the N = 1/2/4/8 decompositions, alone on ONE GPU as a ring of one rank with the real halo
machinery (peer-to-peer: the rank is its own neighbour): what a rank costs before the xGMI hop and the neighbours'
skew.  NOT a multi-GPU measurement.   python tools/strong_scaling_proxy.py"""
import os
import sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402  (before the HIP library)
import advanced_hpc_lbm_amd as L  # noqa: E402
os.environ["LBM_FORCE_EXCHANGE"] = "1"
only = int(sys.argv[1]) if len(sys.argv) > 1 else 0        # e.g. 8192: that width only
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else None   # e.g. "p2p default,rccl": those lines only
for nx, ny, label in ((1024, 1024, "1 GPU"), (1024, 512, "rank of 2"), (1024, 256, "rank of 4"), (1024, 128, "rank of 8"),
                      (8192, 8192, "1 GPU"), (8192, 4096, "rank of 2"), (8192, 2048, "rank of 4"), (8192, 1024, "rank of 8")):
    if only and nx != only:
        continue
    p = L.Param(nx, ny, 1000, 10, 0.1, 0.01, 1.85)
    ob = np.zeros((ny, nx), np.int32)
    ob[:, 0] = ob[:, -1] = 1
    ob[:, nx // 3] = 1
    steps = 4000 if nx == 1024 else 208
    whole = (1024 * 1024 if nx == 1024 else 8192 * 8192) / (nx * ny)
    for mode, name, tb in ((L.EXCHANGE_P2P, "p2p default", 0), (L.EXCHANGE_P2P, "p2p lbm_wave<8>", 8), (L.EXCHANGE_P2P, "p2p lbm_sweep2", 2), (L.EXCHANGE_RCCL, "rccl", 0), (L.EXCHANGE_RCCL, "rccl lbm_sweep2", 2)):
        if (tb == 8 and ny < 32) or (modes and name not in modes):
            continue
        with L.Lattice(p, ob, rank=0, nranks=1, device=0, unique_id=L.rccl_unique_id(), exchange=mode) as lat:
            if tb:
                lat.set_option("time_block", tb)
            lat.run(20)
            lat.run(steps)
            g, w = lat.last_run_ms()
            print(f"{nx}x{ny} ({label}) {name} [steps per pass {int(lat.info('time_block_active'))}]: {g / steps * 1e3:8.2f} us/step  -> x{whole:.0f} slabs = "
                  f"{nx * ny * steps / (g * 1e-3) / 1e6 * whole:9.0f} MLUPS aggregate if perfectly parallel", flush=True)
