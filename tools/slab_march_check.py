#!/usr/bin/env python3
"""Development check: lbm_march across the slabs of one process (neighbour rows read in place) against the undivided
lattice on one GPU (all slabs on device 0).   python tools/slab_march_check.py"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import advanced_hpc_lbm_amd as L  # noqa: E402


def random_case(nx, ny, seed, blocked=0.1):
    rng = np.random.default_rng(seed)
    p = L.Param(nx, ny, 100, 10, 0.1, 0.01, 1.85)
    ob = (rng.random((ny, nx)) < blocked).astype(np.int32)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float32)
    cells = (0.1 * w * (1.0 + 0.2 * (rng.random((ny, nx, 9), dtype=np.float32) - 0.5))).astype(np.float32)
    return p, ob, cells


ok = True
for nx, ny, nslabs, steps, exch in ((256, 64, 2, [4], L.EXCHANGE_COPY), (256, 96, 3, [8, 5], L.EXCHANGE_COPY), (480, 200, 4, [13], L.EXCHANGE_COPY),
                                    (1024, 1024, 8, [16, 3], L.EXCHANGE_COPY), (260, 70, 2, [7], L.EXCHANGE_COPY)):
    p, ob, cells = random_case(nx, ny, 9)
    with L.Lattice(p, ob, cells) as a:
        a.set_option("time_block", 1)
        av_a = np.concatenate([a.run(n) for n in steps])
        st_a = a.read_state()
    with L.Lattice(p, ob, cells, nslabs=nslabs, devices=[0] * nslabs, exchange=exch) as b:
        b.set_option("time_block", 4)
        tb = int(b.info("time_block_active"))
        av_b = np.concatenate([b.run(n) for n in steps])
        st_b = b.read_state()
    same = np.array_equal(st_a.view(np.uint32), st_b.view(np.uint32))
    avok = np.allclose(av_a, av_b, rtol=2e-6, atol=0)
    msg = f"{nx}x{ny} {nslabs} slabs time_block_active {tb} steps {steps}: state {'BIT-EXACT' if same else 'DIFFERS'}, av_vels {'ok' if avok else 'DIFFER'}"
    if not same:
        d = np.argwhere(st_a.view(np.uint32) != st_b.view(np.uint32))
        msg += f"  [{len(d)} differ; first {d[:5].tolist()}; rows {sorted(set(d[:,0].tolist()))[:12]}]"
    print(msg, flush=True)
    ok &= same and avok
print("ALL BIT-EXACT" if ok else "MISMATCHES")
sys.exit(0 if ok else 1)
