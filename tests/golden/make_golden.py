#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ (run in the build container).

Two kinds of fixture, both DATA (inputs + expected outputs), never source:

1. `<deck>.final_state.pressure.f64.npz` for the two decks whose golden
   final_state.dat is absent from the reference mount
   (/root/reference/.MISSING_LARGE_BLOBS: 256x256, 1024x1024).  Produced by the
   double-precision oracle, which reproduces every golden file that IS shipped
   digit for digit (tests/test_oracle_golden.py); the arrays are checked here
   against the sha256 fingerprints recorded in BASELINE.md (addendum), which
   were taken from a double build of the reference itself.  Stored values are
   the pressures as printed with %.12E and parsed back (what check.py would
   load), little-endian float64, row-major jj outer / ii inner.

2. `kat_*.npz` known-answer vectors: small lattices (non-square, random
   obstacles, open top/bottom rows so the y-wrap is live, perturbed initial
   state) advanced 1, 2 and 10 steps by THE REFERENCE ITSELF --
   timestep_new2 from /root/reference/d2q9-bgk.c compiled with strict IEEE
   flags by oracle/Makefile into oracle/_ref/libd2q9_ref_strict.so.  They pin
   the oracle (bit-exact, float) on the GPU box, where the reference is absent.

3. `ref_float_<deck>.npz`: av_vels + final pressure written by the reference
   CLI binary as shipped (float, -Ofast; oracle/_ref/d2q9-bgk) on the small
   decks: the expected float-vs-double deviation our own checker must report.

Usage: python tests/golden/make_golden.py [--pressure 256x256 1024x1024] [--kat] [--ref-float]
"""
import argparse
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import lbm_oracle as O  # noqa: E402

# BASELINE.md addendum: sha256 of the float64-LE pressure arrays
PRESSURE_SHA256 = {
    "256x256": "51d1f8f682c6da7a63b43b997e27fce96c910aaaf44c210b5f8fc1572c55cc2d",
    "1024x1024": "e65843180cc9c608c67d63d1b124e73e613cdd78a625845dc7cf41316e69a00d",
}


def printed(a: np.ndarray) -> np.ndarray:
    """Round-trip through the %.12E text form, as a reader of final_state.dat sees it."""
    return np.array([float("%.12E" % v) for v in a.ravel()], dtype="<f8").reshape(a.shape)


def make_pressure(deck: str) -> None:
    orc = O.Oracle()
    prm = O.read_params(os.path.join(ROOT, f"input_{deck}.params"))
    ob = O.read_obstacles(os.path.join(ROOT, f"obstacles_{deck}.dat"), prm.nx, prm.ny)
    cells = orc.init_cells(prm, np.float64)
    t = time.time()
    av = orc.run(prm, cells, ob, prm.maxIters)
    print(f"{deck}: double oracle ran {prm.maxIters} steps in {time.time() - t:.1f} s", flush=True)
    gold_av = open(os.path.join(HERE, f"{deck}.av_vels.dat")).read()
    assert O.format_av_vels(av) == gold_av, "double oracle does not reproduce the shipped av_vels golden"
    fs = orc.final_state(prm, cells, ob)
    pressure = printed(fs[:, :, 3])
    sha = hashlib.sha256(pressure.tobytes()).hexdigest()
    print(f"{deck}: sha256(pressure f64) = {sha}")
    assert sha == PRESSURE_SHA256[deck], "fingerprint differs from BASELINE.md"
    np.savez_compressed(os.path.join(HERE, f"{deck}.final_state.pressure.f64.npz"),
                        pressure=pressure, reynolds=np.float64(orc.reynolds(prm, cells, ob)))


def make_kat() -> None:
    ref = O.ReferenceStrict()
    rng = np.random.default_rng(20260104)
    cases = [("kat_8x6", 8, 6), ("kat_16x12", 16, 12), ("kat_33x20", 33, 20), ("kat_64x40", 64, 40)]
    for name, nx, ny in cases:
        prm = O.OrcParam(nx, ny, 10, 10, 0.1, 0.005, 1.85)
        rp = O.to_ref_param(prm)
        ob = (rng.random((ny, nx)) < 0.12).astype(np.int32)
        ob[0, 1:-1] = 0          # open bottom and top rows: y-wrap is exercised
        ob[ny - 1, 1:-1] = 0
        ob[ny - 2, : nx // 2] = 0  # accelerate row: half guaranteed fluid, rest random
        w = np.array([4 / 9] + [1 / 9] * 4 + [1 / 36] * 4)
        cells0 = (0.1 * w * (1.0 + 0.2 * (rng.random((ny, nx, 9)) - 0.5))).astype(np.float32)
        # a few cells so thin that the accelerate guard (f3-w1>0 etc.) must refuse them
        cells0[ny - 2, 1, 3] = 1e-6
        cells0[ny - 2, 2, 6] = 1e-7
        out = {"nx": nx, "ny": ny, "reynolds_dim": 10, "density": 0.1, "accel": 0.005, "omega": 1.85,
               "obstacles": ob, "cells0": cells0}
        a, b = cells0.copy(), np.empty_like(cells0)
        av = []
        for tt in range(1, 11):
            av.append(ref.timestep_new2(rp, a, b, ob))
            a, b = b, a
            if tt in (1, 2, 10):
                out[f"cells_after_{tt}"] = a.copy()
        out["av_vels"] = np.array(av, dtype=np.float32)
        out["reynolds_after_10"] = np.float32(ref.calc_reynolds(rp, a, ob))
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(f"{name}: blocked {int(ob.sum())}/{nx * ny}, av_vels[9] = {av[-1]:.9e}")


def make_ref_float(decks) -> None:
    exe = os.path.join(ROOT, "oracle", "_ref", "d2q9-bgk")
    for deck in decks:
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run([exe, os.path.join(ROOT, f"input_{deck}.params"),
                                os.path.join(ROOT, f"obstacles_{deck}.dat")],
                               cwd=td, check=True, capture_output=True, text=True)
            av = O.read_av_vels(os.path.join(td, "av_vels.dat")).astype(np.float32)
            fs = O.read_final_state(os.path.join(td, "final_state.dat"))
            reyn = [ln for ln in r.stdout.splitlines() if ln.startswith("Reynolds")][0].split()[-1]
        ny_nx = deck.split("x")
        nx, ny = int(ny_nx[0]), int(ny_nx[1])
        np.savez_compressed(os.path.join(HERE, f"ref_float_{deck}.npz"), av_vels=av,
                            pressure=fs[:, 5].astype(np.float32).reshape(ny, nx),
                            reynolds=np.float64(reyn))
        print(f"ref_float_{deck}: Reynolds {reyn}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pressure", nargs="*", default=[])
    ap.add_argument("--kat", action="store_true")
    ap.add_argument("--ref-float", nargs="*", default=[])
    args = ap.parse_args()
    O.build()
    for d in args.pressure:
        make_pressure(d)
    if args.kat:
        make_kat()
    if args.ref_float:
        make_ref_float(args.ref_float)
