"""Ad-hoc first-light check on a GPU box (not part of the test suite)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import advanced_hpc_lbm_amd as L
import lbm_oracle as O

print("devices", L.device_count(), flush=True)
orc = O.Oracle()
for deck in ["128x128", "128x256", "256x256", "1024x1024"]:
    p = L.read_params(f"{ROOT}/input_{deck}.params"); ob = L.read_obstacles(f"{ROOT}/obstacles_{deck}.dat", p)
    op = O.read_params(f"{ROOT}/input_{deck}.params")
    n = 50
    cells = orc.init_cells(op, np.float32)
    av_o = orc.run(op, cells, ob, n)
    for V in (4, 2, 1):
        with L.Lattice(p, ob) as lat:
            lat.set_option("vector_width", V)
            av = lat.run(n)
            st = lat.read_state()
        err = np.abs(st - cells).max() / np.abs(cells).max()
        print(deck, "V", V, "state rel err", err, "av rel err", np.abs(av - av_o).max() / av_o.max(), flush=True)
# full runs against goldens
sys.path.insert(0, os.path.join(ROOT, "check"))
import check_results as CR
for deck in ["128x128", "128x256", "256x256", "1024x1024"]:
    p = L.read_params(f"{ROOT}/input_{deck}.params"); ob = L.read_obstacles(f"{ROOT}/obstacles_{deck}.dat", p)
    with L.Lattice(p, ob) as lat:
        t = time.time(); av = lat.run(p.maxIters); wall = time.time() - t
        g, w = lat.last_run_ms()
        fs = lat.final_state(); re = lat.reynolds()
    gold = np.loadtxt(f"{ROOT}/tests/golden/{deck}.av_vels.dat", usecols=[1])
    dev = CR.worst_deviation(gold, av.astype(np.float64))
    mlups = p.nx * p.ny * p.maxIters / (g * 1e-3) / 1e6
    print(f"{deck}: {p.maxIters} steps gpu {g:.1f} ms wall {wall*1e3:.1f} ms -> {mlups:.0f} MLUPS ({mlups*72/1e3:.0f} GB/s); "
          f"av_vels worst {dev['percent']:.4f}% at {dev['index']}; Reynolds {re:.9e}", flush=True)
# synthetic 8192^2
p = L.Param(8192, 8192, 100, 10, 0.1, 0.01, 1.85)
ob = np.zeros((8192, 8192), np.int32); ob[0, :] = ob[-1, :] = 1; ob[:, 0] = ob[:, -1] = 1; ob[:, 2730] = 1
for V in (4, 2):
    with L.Lattice(p, ob) as lat:
        lat.set_option("vector_width", V)
        lat.run(5)
        for rep in range(3):
            av = lat.run(100); g, w = lat.last_run_ms()
            mlups = 8192 * 8192 * 100 / (g * 1e-3) / 1e6
            print(f"8192^2 V={V}: {g/100*1e3:.1f} us/step {mlups:.0f} MLUPS {mlups*72/1e3:.0f} GB/s frac {mlups*72/8e6:.3f}", flush=True)
