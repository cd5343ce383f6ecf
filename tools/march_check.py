#!/usr/bin/env python3
"""Development check of the marching kernel (lbm_march, time_block = 4) on one GPU: bit-identity with
the one-step kernel on random lattices (partial strips, ragged chunks, step counts with remainders),
then us/step against lbm_sweep2 on big lattices.   python tools/march_check.py [quick]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import advanced_hpc_lbm_amd as L  # noqa: E402
from make_deck import obstacle_map  # noqa: E402


def random_case(nx, ny, seed, blocked=0.1):
    rng = np.random.default_rng(seed)
    p = L.Param(nx, ny, 100, 10, 0.1, 0.01, 1.85)
    ob = (rng.random((ny, nx)) < blocked).astype(np.int32)
    w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4, dtype=np.float32)
    cells = (0.1 * w * (1.0 + 0.2 * (rng.random((ny, nx, 9), dtype=np.float32) - 0.5))).astype(np.float32)
    return p, ob, cells


def compare(nx, ny, rows, steps_list, seed=1, tb=4, kernel=0):
    p, ob, cells = random_case(nx, ny, seed)
    with L.Lattice(p, ob, cells) as a:
        a.set_option("time_block", 1)
        av_a = np.concatenate([a.run(n) for n in steps_list])
        st_a = a.read_state()
    with L.Lattice(p, ob, cells) as b:
        b.set_option("march_kernel", kernel)
        b.set_option("time_block", tb)
        if rows:
            b.set_option("wave_rows" if kernel else "march_rows", rows)
        assert int(b.info("time_block_active")) == tb, "marching kernel not eligible"
        av_b = np.concatenate([b.run(n) for n in steps_list])
        st_b = b.read_state()
        rows_used = int(b.info("wave_rows" if kernel else "march_rows"))
    same = np.array_equal(st_a.view(np.uint32), st_b.view(np.uint32))
    avok = np.allclose(av_a, av_b, rtol=2e-6, atol=0)
    msg = f"{'lbm_wave' if kernel else 'lbm_march'}<{tb}> {nx}x{ny} rows/chunk {rows_used} steps {steps_list}: state {'BIT-EXACT' if same else 'DIFFERS'}, av_vels {'ok' if avok else 'DIFFER'}"
    if not same:
        d = np.argwhere(st_a.view(np.uint32) != st_b.view(np.uint32))
        msg += f"  [{len(d)} values differ; first (y,x,k) = {d[:5].tolist()}; planes {sorted(set(d[:, 2].tolist()))}; " \
               f"x range {d[:, 1].min()}..{d[:, 1].max()} y range {d[:, 0].min()}..{d[:, 0].max()}]"
    if not avok:
        bad = np.argwhere(~np.isclose(av_a, av_b, rtol=2e-6, atol=0)).ravel()
        msg += f"  [av steps off: {bad[:8].tolist()} rel {np.max(np.abs(av_a - av_b) / np.abs(av_a)):.2e}]"
    print(msg, flush=True)
    return same and avok


def timing(n, steps, tbs=(2, 4), rows=None, kernel=0, form=0):
    p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85)
    ob = obstacle_map(n, n)
    for tb in tbs:
        with L.Lattice(p, ob) as lat:
            lat.set_option("march_kernel", kernel)
            lat.set_option("time_block", tb)
            if tb >= 4 and rows:
                lat.set_option("wave_rows" if kernel else "march_rows", rows)
            lat.run(steps)
            best = 1e9
            for _ in range(3):
                lat.run(steps)
                g, w = lat.last_run_ms()
                best = min(best, g)
            print(f"{n}x{n} time_block {tb} kernel {kernel} form {form} (active {int(lat.info('time_block_active'))}, rows {int(lat.info('wave_rows' if kernel else 'march_rows'))}, "
                  f"capacity {int(lat.info('wave_capacity'))}): "
                  f"{best * 1e3 / steps:.2f} us/step, {n * n * steps / best / 1e6:.1f} GLUPS", flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ok = True
    cases = [
        (256, 64, 0, [4]), (256, 64, 16, [4]), (256, 64, 7, [8, 5]), (260, 40, 0, [4]), (448, 100, 33, [12]),
        (480, 70, 0, [13]), (1000, 24, 8, [9]), (1024, 1024, 0, [16, 3]), (2048, 512, 0, [8]),
    ]
    for nx, ny, rows, steps in cases:
        try:
            ok &= compare(nx, ny, rows, steps)
        except (L.LbmError, AssertionError) as e:
            print(f"{nx}x{ny} rows {rows}: ERROR {e}", flush=True)
            ok = False
    wcases = [(64, 40, 0, [4]), (100, 30, 7, [8, 5]), (130, 77, 16, [13])] + cases[4:]
    for tb in (4, 6, 8):
        for nx, ny, rows, steps in wcases:
            steps = [n * tb // 4 + (1 if n % 4 else 0) for n in steps]
            try:
                ok &= compare(nx, ny, rows, steps, tb=tb, kernel=1)
            except (L.LbmError, AssertionError) as e:
                print(f"lbm_wave<{tb}> {nx}x{ny} rows {rows}: ERROR {e}", flush=True)
                ok = False
    print("ALL BIT-EXACT" if ok else "MISMATCHES", flush=True)
    if not quick:
        timing(8192, 96, tbs=(2, 4))
        for rows in (24, 32, 48, 64, 96):
            timing(8192, 96, tbs=(4, 6, 8), rows=rows, kernel=1)
        for rows in (32, 64):
            timing(8192, 96, tbs=(4, 6, 8), rows=rows, kernel=1, form=1)
        for rows in (32, 64):
            timing(4096, 192, tbs=(4, 6, 8), rows=rows, kernel=1)
            timing(2048, 384, tbs=(4, 6, 8), rows=rows, kernel=1)
            timing(1024, 1920, tbs=(4, 6, 8), rows=rows, kernel=1)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
