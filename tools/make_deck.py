#!/usr/bin/env python3
"""Writes a synthetic input deck in the reference's two file formats.

    python tools/make_deck.py NX NY [--iters N] [--outdir DIR] [--porous FRACTION --seed S]

Produces  input_NXxNY.params   seven lines: nx ny maxIters reynolds_dim density accel omega
                               (read by initialise(), /root/reference/d2q9-bgk.c:2736-2762)
          obstacles_NXxNY.dat  one "x y 1" line per blocked cell
                               (d2q9-bgk.c:2844-2857; duplicates are allowed, none are written)

Geometry = the shipped 1024x1024 deck scaled to NX x NY: closed box
(rows 0 and NY-1, columns 0 and NX-1) plus a full-height wall at x = 341*NX/1024 (2730 for
NX = 8192); density 0.1, accel 0.01, omega 1.85, reynolds_dim 10.  --porous adds uniformly random
blocked cells (a PCG64 stream seeded with --seed) to the interior.
bench.py builds the same obstacle map in memory (synthetic_obstacles); tests compare the two.
"""
import argparse
import os

import numpy as np


def wall_x(nx: int) -> int:
    return 2730 if nx == 8192 else (341 * nx) // 1024


def obstacle_map(nx: int, ny: int, porous: float = 0.0, seed: int = 12345) -> np.ndarray:
    ob = np.zeros((ny, nx), dtype=np.int32)
    ob[0, :] = ob[-1, :] = 1
    ob[:, 0] = ob[:, -1] = 1
    ob[:, wall_x(nx)] = 1
    if porous > 0.0:
        rng = np.random.Generator(np.random.PCG64(seed))
        ob[1:-1, 1:-1] |= (rng.random((ny - 2, nx - 2)) < porous).astype(np.int32)
    return ob


def write_deck(nx, ny, iters, outdir=".", porous=0.0, seed=12345, density=0.1, accel=0.01, omega=1.85,
               reynolds_dim=10):
    os.makedirs(outdir, exist_ok=True)
    pf = os.path.join(outdir, f"input_{nx}x{ny}.params")
    of = os.path.join(outdir, f"obstacles_{nx}x{ny}.dat")
    with open(pf, "w") as f:
        f.write(f"{nx}\n{ny}\n{iters}\n{reynolds_dim}\n{density:g}\n{accel:g}\n{omega:g}\n")
    ob = obstacle_map(nx, ny, porous, seed)
    ys, xs = np.nonzero(ob)                       # row-major: y outer, x inner
    with open(of, "w") as f:
        f.write("".join(f"{x} {y} 1\n" for x, y in zip(xs.tolist(), ys.tolist())))
    return pf, of, int(ob.sum())


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("nx", type=int)
    ap.add_argument("ny", type=int)
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--porous", type=float, default=0.0)
    ap.add_argument("--seed", type=int, default=12345)
    a = ap.parse_args()
    if a.nx < 4 or a.ny < 4:
        raise SystemExit("need at least 4 x 4 cells for a box with a wall")
    pf, of, nb = write_deck(a.nx, a.ny, a.iters, a.outdir, a.porous, a.seed)
    print(f"{pf}\n{of}\n{nb} blocked cells")


if __name__ == "__main__":
    main()
