#!/usr/bin/env python3
"""One rank's 8192 x 1024 slab as a ring of one rank under RCCL ghost bands (or peer-to-peer), a few groups: for
rocprofv3 --kernel-trace timelines.   python3 tools/band_trace.py [rccl|p2p] [steps] [ny]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
import advanced_hpc_lbm_amd as L  # noqa: E402
os.environ["LBM_FORCE_EXCHANGE"] = "1"
mode = L.EXCHANGE_P2P if (len(sys.argv) > 1 and sys.argv[1] == "p2p") else L.EXCHANGE_RCCL
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
ny = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
nx = 8192
p = L.Param(nx, ny, 1000, 10, 0.1, 0.01, 1.85)
ob = np.zeros((ny, nx), np.int32)
ob[:, 0] = ob[:, -1] = 1
ob[:, nx // 3] = 1
with L.Lattice(p, ob, rank=0, nranks=1, device=0, unique_id=L.rccl_unique_id(), exchange=mode) as lat:
    lat.run(16)
    lat.run(steps)
    g, w = lat.last_run_ms()
    print(f"{nx}x{ny} mode {sys.argv[1] if len(sys.argv) > 1 else 'rccl'} [steps per pass {int(lat.info('time_block_active'))}]: {g / steps * 1e3:.2f} us/step (gpu), {w / steps * 1e3:.2f} (wall)")
