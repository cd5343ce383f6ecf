#!/usr/bin/env python3
"""Development check of the resident kernel (lbm_resident) on one GPU:
bit-identity with the one-step streaming kernel on random lattices of several shapes / tilings,
then us/step of both engines on the shipped decks.   python tools/resident_check.py [quick]"""
import os
import sys
import time

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


def compare(nx, ny, tile, steps_list, seed=1):
    p, ob, cells = random_case(nx, ny, seed)
    with L.Lattice(p, ob, cells) as a:
        a.set_option("engine", 1)
        a.set_option("time_block", 1)
        av_a = np.concatenate([a.run(n) for n in steps_list])
        st_a = a.read_state()
    with L.Lattice(p, ob, cells) as b:
        b.set_option("engine", 2)
        if tile is not None:
            tx, ty, v = tile
            b.set_option("resident_tile", tx * 100000 + ty * 10 + v)
        got = int(b.info("resident_tile"))
        av_b = np.concatenate([b.run(n) for n in steps_list])
        assert int(b.info("engine_last")) == 2
        st_b = b.read_state()
    same = np.array_equal(st_a.view(np.uint32), st_b.view(np.uint32))
    avok = np.allclose(av_a, av_b, rtol=2e-6, atol=0)
    msg = f"{nx}x{ny} tile {got // 100000}x{(got // 10) % 10000} v{got % 10} steps {steps_list}: state {'BIT-EXACT' if same else 'DIFFERS'}, av_vels {'ok' if avok else 'DIFFER'}"
    if not same:
        d = np.argwhere(st_a.view(np.uint32) != st_b.view(np.uint32))
        msg += f"  [{len(d)} values differ; first (y,x,k) = {d[:6].tolist()}; planes {sorted(set(d[:, 2].tolist()))}; " \
               f"x mod tx {sorted(set((d[:, 1] % (got // 100000)).tolist()))[:8]} y mod ty {sorted(set((d[:, 0] % ((got // 10) % 10000)).tolist()))[:8]}]"
    if not avok:
        msg += f"  [av max rel {np.max(np.abs(av_a - av_b) / np.abs(av_a)):.2e}]"
    print(msg, flush=True)
    return same and avok


def timing(deck, steps):
    pf, of = os.path.join(ROOT, f"input_{deck}.params"), os.path.join(ROOT, f"obstacles_{deck}.dat")
    p = L.read_params(pf)
    ob = L.read_obstacles(of, p)
    for eng in (1, 2):
        with L.Lattice(p, ob) as lat:
            lat.set_option("engine", eng)
            lat.run(steps)
            t0 = time.perf_counter()
            lat.run(steps)
            dt = time.perf_counter() - t0
            g, w = lat.last_run_ms()
            print(f"{deck} engine {eng} (last {int(lat.info('engine_last'))}, tile {int(lat.info('resident_tile'))}): "
                  f"{dt * 1e6 / steps:.3f} us/step wall, {g * 1e3 / steps:.3f} us/step gpu, "
                  f"{p.nx * p.ny * steps / dt / 1e9:.1f} GLUPS", flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ok = True
    cases = [
        (64, 64, (16, 16, 1), [1]), (64, 64, (16, 16, 1), [2]), (64, 64, (16, 16, 1), [7, 4]),
        (64, 64, (16, 16, 4), [9]), (64, 64, (32, 32, 2), [9]), (64, 64, (64, 64, 4), [9]),   # one tile: its own neighbour
        (128, 128, None, [11, 2]), (128, 256, None, [12]), (256, 256, None, [12]),
        (96, 80, (32, 16, 4), [10]), (96, 80, (8, 5, 1), [10]), (64, 48, (4, 1, 4), [6]), (40, 36, (8, 12, 2), [10]),
        (1024, 1024, None, [21]),
    ]
    for nx, ny, tile, steps in cases:
        try:
            ok &= compare(nx, ny, tile, steps)
        except L.LbmError as e:
            print(f"{nx}x{ny} tile {tile}: ERROR {e}", flush=True)
            ok = False
    print("ALL BIT-EXACT" if ok else "MISMATCHES", flush=True)
    if not quick:
        for deck, steps in (("128x128", 2000), ("128x256", 2000), ("256x256", 2000), ("1024x1024", 2000)):
            timing(deck, steps)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
