#!/usr/bin/env python3
"""us/step of lbm_march and lbm_wave<K> over rows-per-chunk on a generated deck: python tools/wave_sweep.py N [K ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import advanced_hpc_lbm_amd as L  # noqa: E402
from make_deck import obstacle_map  # noqa: E402

n = int(sys.argv[1])
ks = [int(v) for v in sys.argv[2:]] or [8]
p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85)
ob = obstacle_map(n, n)
steps = max(24, (1 << 28) // (n * n) * 24 // 24 * 24 // 4)


def run(tb, kernel, rows):
    with L.Lattice(p, ob) as lat:
        lat.set_option("march_kernel", kernel)
        lat.set_option("time_block", tb)
        if rows:
            lat.set_option("wave_rows" if kernel else "march_rows", rows)
        nst = steps // tb * tb
        lat.run(nst)
        best = 1e9
        for _ in range(3):
            lat.run(nst)
            best = min(best, lat.last_run_ms()[0])
        name = ("lbm_wave<%d>" % tb) if kernel else "lbm_march"
        print(f"{n}x{n} {name} rows {rows or int(lat.info('wave_rows' if kernel else 'march_rows'))}: "
              f"{best * 1e3 / nst:.2f} us/step, {n * n * nst / best / 1e6:.1f} GLUPS", flush=True)


run(4, 0, 0)
for k in ks:
    for rows in (32, 48, 64, 96, 128, 192, 256):
        if rows <= n:
            run(k, 1, rows)
