import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import advanced_hpc_lbm_amd as L
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for deck, steps in (("128x128", 20000), ("256x256", 20000), ("1024x1024", 8000)):
    p = L.read_params(f"{ROOT}/input_{deck}.params"); ob = L.read_obstacles(f"{ROOT}/obstacles_{deck}.dat", p)
    for nt in (256, 512, 1024, 256, 512, 1024):
        with L.Lattice(p, ob) as lat:
            lat.set_option("t2_threads", nt)
            lat.run(100); lat.run(steps); g, w = lat.last_run_ms()
            print(f"{deck} no-exchange t2_threads={nt}: {g/steps*1e3:7.2f} us/step {p.nx*p.ny*steps/(g*1e-3)/1e6:9.0f} MLUPS", flush=True)
n = 8192
p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85); ob = np.zeros((n, n), np.int32); ob[:, 0] = ob[:, -1] = 1; ob[0, :] = ob[-1, :] = 1; ob[:, 2730] = 1
for nt in (256, 512, 256, 512):
    with L.Lattice(p, ob) as lat:
        lat.set_option("t2_threads", nt)
        lat.run(10); lat.run(200); g, w = lat.last_run_ms()
        print(f"8192 no-exchange t2_threads={nt}: {g/200*1e3:7.2f} us/step {n*n*200/(g*1e-3)/1e6:9.0f} MLUPS", flush=True)
