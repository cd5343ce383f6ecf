#!/usr/bin/env python3
"""us/step of lbm_march and of lbm_wave<K> (one or two columns per lane) over rows per chunk on a generated deck.
    python tools/wave_sweep.py N [--k 8 6] [--cols 1 2] [--rows 64 96 128 149 192] [--no-march]"""
import argparse
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import advanced_hpc_lbm_amd as L  # noqa: E402
from make_deck import obstacle_map  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("n", type=int)
ap.add_argument("--ny", type=int, default=0)
ap.add_argument("--k", type=int, nargs="*", default=[8])
ap.add_argument("--cols", type=int, nargs="*", default=[1])
ap.add_argument("--rows", type=int, nargs="*", default=[32, 48, 64, 96, 128, 192, 256])
ap.add_argument("--no-march", action="store_true")
args = ap.parse_args()
n, ny = args.n, args.ny or args.n
p = L.Param(n, ny, 1000, 10, 0.1, 0.01, 1.85)
ob = obstacle_map(n, ny)
steps = max(24, (1 << 28) // (n * ny) * 24 // 24 * 24 // 4)


def run(tb, kernel, rows, cols=1):
    with L.Lattice(p, ob) as lat:
        lat.set_option("march_kernel", kernel)
        lat.set_option("time_block", tb)
        if kernel:
            lat.set_option("wave_cols", cols)
        if rows:
            lat.set_option("wave_rows" if kernel else "march_rows", rows)
        nst = steps // tb * tb
        lat.run(nst)
        best = 1e9
        for _ in range(3):
            lat.run(nst)
            best = min(best, lat.last_run_ms()[0])
        name = ("lbm_wave<%d> x%d" % (tb, int(lat.info("wave_cols_active")))) if kernel else "lbm_march"
        waves = ""
        if kernel:
            oc = int(lat.info("wave_out_cols"))
            r = rows or int(lat.info("wave_rows"))
            nw = -(-n // oc) * -(-ny // r)
            waves = f", {nw} waves / {int(lat.info('wave_capacity'))} slots"
        print(f"{n}x{ny} {name} rows {rows or int(lat.info('wave_rows' if kernel else 'march_rows'))}: "
              f"{best * 1e3 / nst:.2f} us/step, {n * ny * nst / best / 1e6:.1f} GLUPS{waves}", flush=True)


if not args.no_march:
    run(4, 0, 0)
for k in args.k:
    for cols in args.cols:
        for rows in args.rows:
            if rows <= ny:
                run(k, 1, rows, cols)
