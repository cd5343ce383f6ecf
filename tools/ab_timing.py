#!/usr/bin/env python3
"""A/B timing of the main kernels (development): python tools/ab_timing.py <label>"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import advanced_hpc_lbm_amd as L  # noqa: E402
from make_deck import obstacle_map  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else ""


def run(n, steps, tb, kernel=0, rows=0):
    p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85)
    with L.Lattice(p, obstacle_map(n, n)) as lat:
        lat.set_option("march_kernel", kernel)
        lat.set_option("time_block", tb)
        if rows:
            lat.set_option("wave_rows" if kernel else "march_rows", rows)
        lat.run(steps)
        best = 1e9
        for _ in range(3):
            lat.run(steps)
            best = min(best, lat.last_run_ms()[0])
        name = {1: "lbm_sweep", 2: "lbm_sweep2"}.get(tb, ("lbm_wave<%d>" % tb) if kernel else "lbm_march")
        print(f"[{label}] {n}x{n} {name} rows {rows}: {best * 1e3 / steps:.2f} us/step, {n * n * steps / best / 1e6:.1f} GLUPS", flush=True)


run(1024, 1920, 2)
run(8192, 96, 1)
run(8192, 96, 2)
run(8192, 96, 4, 0)
for k in (4, 6, 8):
    run(8192, 96, k, 1, 64)
run(8192, 96, 6, 1, 48)
run(4096, 192, 6, 1, 64)
