#!/usr/bin/env python3
"""Development check of the register-tile engine (lbm_regtile, engine 3) on one GPU: bit-identity with the
one-step streaming kernel on random lattices / tilings, then us/step against lbm_sweep2 on the shipped decks.
   python tools/regtile_check.py [quick]"""
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


def compare(nx, ny, tile, steps_list, seed=1):
    p, ob, cells = random_case(nx, ny, seed)
    with L.Lattice(p, ob, cells) as a:
        a.set_option("time_block", 1)
        av_a = np.concatenate([a.run(n) for n in steps_list])
        st_a = a.read_state()
    with L.Lattice(p, ob, cells) as b:
        if tile is not None:
            b.set_option("regtile", tile[0] * 10 + tile[1])
        b.set_option("engine", 3)
        got = int(b.info("regtile"))
        av_b = np.concatenate([b.run(n) for n in steps_list])
        assert int(b.info("engine_last")) == 3
        st_b = b.read_state()
    same = np.array_equal(st_a.view(np.uint32), st_b.view(np.uint32))
    avok = np.allclose(av_a, av_b, rtol=2e-6, atol=0)
    ty = got // 10
    msg = f"{nx}x{ny} tile 64x{ty} ({got % 10} rows per wave) steps {steps_list}: state {'BIT-EXACT' if same else 'DIFFERS'}, av_vels {'ok' if avok else 'DIFFER'}"
    if not same:
        d = np.argwhere(st_a.view(np.uint32) != st_b.view(np.uint32))
        msg += f"  [{len(d)} values differ; first (y,x,k) = {d[:6].tolist()}; planes {sorted(set(d[:, 2].tolist()))}; " \
               f"x mod 64 {sorted(set((d[:, 1] % 64).tolist()))[:10]} y mod ty {sorted(set((d[:, 0] % ty).tolist()))[:10]}]"
    if not avok:
        msg += f"  [av max rel {np.max(np.abs(av_a - av_b) / np.abs(av_a)):.2e}]"
    print(msg, flush=True)
    return same and avok


def timing(deck, steps, tiles=(None,)):
    pf, of = os.path.join(ROOT, f"input_{deck}.params"), os.path.join(ROOT, f"obstacles_{deck}.dat")
    p = L.read_params(pf)
    ob = L.read_obstacles(of, p)
    with L.Lattice(p, ob) as lat:
        lat.run(steps)
        best = min(lat.run(steps) is None or lat.last_run_ms()[0] for _ in range(3))
        print(f"{deck} streaming (time_block {int(lat.info('time_block_active'))}): {best * 1e3 / steps:.3f} us/step, {p.nx * p.ny * steps / best / 1e6:.1f} GLUPS", flush=True)
    for tile in tiles:
        with L.Lattice(p, ob) as lat:
            try:
                if tile:
                    lat.set_option("regtile", tile[0] * 10 + tile[1])
                lat.set_option("engine", 3)
            except L.LbmError as e:
                print(deck, tile, "not usable:", e)
                continue
            lat.run(steps)
            best = min(lat.run(steps) is None or lat.last_run_ms()[0] for _ in range(3))
            print(f"{deck} lbm_regtile {int(lat.info('regtile'))}: {best * 1e3 / steps:.3f} us/step, {p.nx * p.ny * steps / best / 1e6:.1f} GLUPS", flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    ok = True
    cases = [
        (64, 4, (4, 4), [1]), (64, 4, (4, 4), [2]), (64, 8, (4, 4), [5]), (64, 8, (8, 4), [5, 2]), (128, 16, (4, 2), [7]),
        (128, 16, (16, 1), [7]), (128, 128, None, [11, 2]), (128, 256, None, [12]), (256, 256, None, [12]), (256, 256, (8, 4), [9]),
        (192, 96, (32, 4), [10]), (1024, 1024, None, [21]),
    ]
    for nx, ny, tile, steps in cases:
        try:
            ok &= compare(nx, ny, tile, steps)
        except (L.LbmError, AssertionError) as e:
            print(f"{nx}x{ny} tile {tile}: ERROR {e}", flush=True)
            ok = False
    print("ALL BIT-EXACT" if ok else "MISMATCHES", flush=True)
    if not quick:
        timing("1024x1024", 2000)
        timing("256x256", 4000, tiles=(None, (8, 4), (16, 4), (4, 2)))
        timing("128x128", 4000, tiles=(None, (4, 4), (8, 4), (2, 2)))
        timing("128x256", 4000)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
