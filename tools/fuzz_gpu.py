"""Randomised cross-check on a GPU box (development aid, not part of the suite): random lattice sizes,
obstacle densities, step splits and decompositions; every configuration must leave the lattice
bit-identical to the plain single-step kernel on one slab."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import advanced_hpc_lbm_amd as L

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
t0, n, bad = time.time(), 0, 0
n_reg, n_reg_slabs = 0, 0       # configurations that ran lbm_regtile: alone / across slabs
w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)
while time.time() - t0 < budget:
    nx = int(rng.choice([64, 128, 192, 256, 65, 100, 130, 66, 320, 1000]))
    ny = int(rng.choice([16, 32, 48, 64, 96, 17, 37, 50, 128, 600]))
    p = L.Param(nx, ny, 10, 3, 0.1, float(rng.choice([0.005, 0.02, 0.05])), float(rng.choice([1.0, 1.7, 1.85])))
    ob = (rng.random((ny, nx)) < rng.choice([0.0, 0.05, 0.3])).astype(np.int32)
    c0 = (w * 0.1 * (1 + 0.3 * (rng.random((ny, nx, 9)) - 0.5))).astype(np.float32)
    splits = [int(v) for v in rng.integers(1, 9, size=int(rng.integers(1, 4)))]
    with L.Lattice(p, ob, c0) as lat:
        lat.set_option("time_block", 1)
        av_ref = np.concatenate([lat.run(k) for k in splits]); ref = lat.read_state()
    configs = [dict(nslabs=1)]
    for ns in (2, 3, 4):
        if ny // ns >= 2:
            configs += [dict(nslabs=ns, exchange=L.EXCHANGE_COPY), dict(nslabs=ns, exchange=L.EXCHANGE_P2P)]
    for cfg in configs:
        for threads in (None, 256, 512, 1024):       # None: the default engine (lbm_regtile, alone or across slabs, where a size tiles)
            ns = cfg["nslabs"]
            kw = dict(cfg); kw["devices"] = [0] * ns
            with L.Lattice(p, ob, c0, **kw) as lat:
                if threads is not None:
                    lat.set_option("t2_threads", threads)
                av = np.concatenate([lat.run(k) for k in splits]); st = lat.read_state()
                tb = int(lat.info("time_block_active"))
                if int(lat.info("engine_last")) == 3:
                    n_reg += ns == 1
                    n_reg_slabs += ns > 1
            n += 1
            if not np.array_equal(st.view(np.uint32), ref.view(np.uint32)) or not np.allclose(av, av_ref, rtol=5e-6, atol=0):
                bad += 1
                print("MISMATCH", nx, ny, cfg, threads, splits, "tb", tb, "max diff", float(np.abs(st - ref).max()), flush=True)
print(f"fuzz: {n} configurations ({n_reg} through lbm_regtile alone, {n_reg_slabs} through register tiles across slabs), {bad} mismatches, {time.time() - t0:.0f} s")
