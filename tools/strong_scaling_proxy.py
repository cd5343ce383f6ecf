import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import advanced_hpc_lbm_amd as L
os.environ["LBM_FORCE_EXCHANGE"] = "1"
for nx, ny, label in ((1024, 1024, "1 GPU"), (1024, 512, "rank of 2"), (1024, 256, "rank of 4"), (1024, 128, "rank of 8"),
                      (8192, 8192, "1 GPU"), (8192, 4096, "rank of 2"), (8192, 2048, "rank of 4"), (8192, 1024, "rank of 8")):
    p = L.Param(nx, ny, 1000, 10, 0.1, 0.01, 1.85)
    ob = np.zeros((ny, nx), np.int32); ob[:, 0] = ob[:, -1] = 1; ob[:, nx // 3] = 1
    steps = 4000 if nx == 1024 else 200
    for mode, name, nt in ((L.EXCHANGE_P2P, "p2p", 256), (L.EXCHANGE_P2P, "p2p 512t", 512), (L.EXCHANGE_P2P, "p2p 1024t", 1024), (L.EXCHANGE_RCCL, "rccl", 256)):
        with L.Lattice(p, ob, rank=0, nranks=1, device=0, unique_id=L.rccl_unique_id(), exchange=mode) as lat:
            lat.set_option("t2_threads", nt)
            lat.run(20); lat.run(steps); g, w = lat.last_run_ms()
            print(f"{nx}x{ny} ({label}) {name}: {g/steps*1e3:8.2f} us/step  -> x{(1024*1024 if nx==1024 else 8192*8192)/(nx*ny)} slabs = {nx*ny*steps/(g*1e-3)/1e6*((1024*1024 if nx==1024 else 8192*8192)/(nx*ny)):9.0f} MLUPS aggregate if perfectly parallel", flush=True)
