import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import advanced_hpc_lbm_amd as L
n = 16384
p = L.Param(n, n, 6, 10, 0.1, 0.01, 1.85)
ob = np.zeros((n, n), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1; ob[:, 5461] = 1
rng = np.random.default_rng(5); ob[rng.integers(1, n - 1, 200000), rng.integers(1, n - 1, 200000)] = 1
t = time.time()
with L.Lattice(p, ob) as lat:
    m0 = lat.total_density(); av1 = lat.run(6); g, w = lat.last_run_ms(); m1 = lat.total_density(); f1 = lat.final_state()
print("16384^2: 6 steps gpu ms", g, "MLUPS", n * n * 6 / (g * 1e-3) / 1e6, "mass drift", abs(m1 - m0) / m0, flush=True)
with L.Lattice(p, np.roll(ob, 7, axis=1)) as lat:
    av2 = lat.run(6); f2 = lat.final_state()
ok = np.array_equal(np.roll(f1, 7, axis=1).view(np.uint32), f2.view(np.uint32))
print("translation invariance bit-exact:", ok, "av close:", np.allclose(av1, av2, rtol=2e-6), "total s", time.time() - t, flush=True)
with L.Lattice(p, ob) as lat:
    lat.set_option("time_block", 1); av3 = lat.run(6); f3 = lat.final_state()
print("two-step == one-step bit-exact:", np.array_equal(f1.view(np.uint32), f3.view(np.uint32)), flush=True)
# the default from ~8192^2 up: eight steps per pass in registers (lbm_wave<8>); 19 steps = two passes + a pair + one
with L.Lattice(p, ob) as lat:
    av4 = lat.run(19); g4, _ = lat.last_run_ms(); tb = int(lat.info("time_block_active")); s4 = lat.read_state()
with L.Lattice(p, ob) as lat:
    lat.set_option("time_block", 1); av5 = lat.run(19); s5 = lat.read_state()
print(f"default ({tb} steps per pass) == one-step bit-exact over 19 steps:", np.array_equal(s4.view(np.uint32), s5.view(np.uint32)),
      "av close:", np.allclose(av4, av5, rtol=2e-6), "MLUPS", n * n * 19 / (g4 * 1e-3) / 1e6, flush=True)
