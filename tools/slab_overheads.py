"""How much the slab machinery costs on ONE GPU (development measurement, not a test):
single slab vs forced RCCL self-ring vs several slabs with peer-copy halos."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
import advanced_hpc_lbm_amd as L

def wl(n):
    if n == 1024:
        p = L.read_params(f"{ROOT}/input_1024x1024.params"); ob = L.read_obstacles(f"{ROOT}/obstacles_1024x1024.dat", p)
    else:
        p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85)
        ob = np.zeros((n, n), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1; ob[:, 2730 if n == 8192 else n // 3] = 1
    return p, ob

def run(tag, p, ob, steps, **kw):
    tb = kw.pop("time_block", 2)
    with L.Lattice(p, ob, **kw) as lat:
        lat.set_option("time_block", tb)
        lat.run(20)
        lat.run(steps); g, w = lat.last_run_ms()
        mlups = p.nx * p.ny * steps / (g * 1e-3) / 1e6
        print(f"{tag:44s} tb_active={int(lat.info('time_block_active'))} {g/steps*1e3:9.2f} us/step  {mlups:9.0f} MLUPS  wall/gpu {w/g:.2f}", flush=True)

for n, steps in ((1024, 4000), (8192, 200)):
    p, ob = wl(n)
    print(f"--- {n}x{n}")
    run("1 slab, no exchange", p, ob, steps)
    run("1 slab, no exchange, time_block 1", p, ob, steps, time_block=1)
    os.environ["LBM_FORCE_EXCHANGE"] = "1"
    run("rank mode, RCCL self-ring", p, ob, steps, rank=0, nranks=1, device=0, unique_id=L.rccl_unique_id())
    run("rank mode, RCCL self-ring, time_block 1", p, ob, steps, rank=0, nranks=1, device=0, unique_id=L.rccl_unique_id(), time_block=1)
    run("1 slab, COPY self-ring", p, ob, steps, nslabs=1, devices=[0], exchange=L.EXCHANGE_COPY)
    run("1 slab, P2P self-ring", p, ob, steps, nslabs=1, devices=[0], exchange=L.EXCHANGE_P2P)
    run("1 slab, P2P self-ring, time_block 1", p, ob, steps, nslabs=1, devices=[0], exchange=L.EXCHANGE_P2P, time_block=1)
    del os.environ["LBM_FORCE_EXCHANGE"]
    for ns in (2, 4):
        run(f"{ns} slabs on one GPU, P2P", p, ob, steps, nslabs=ns, devices=[0] * ns, exchange=L.EXCHANGE_P2P)
    for ns in (2, 8):
        run(f"{ns} slabs on one GPU, COPY", p, ob, steps, nslabs=ns, devices=[0] * ns, exchange=L.EXCHANGE_COPY)
        run(f"{ns} slabs on one GPU, COPY, time_block 1", p, ob, steps, nslabs=ns, devices=[0] * ns, exchange=L.EXCHANGE_COPY, time_block=1)
