#!/usr/bin/env python3
"""Golden-file comparison for d2q9-bgk outputs (own implementation).

Semantics follow the reference's checker (/root/reference/check/check.py,
SURVEY.md Appendix C) so that a run accepted here is accepted there:

  * av_vels.dat: the value after "N:" on every line; final_state.dat: columns
    0, 1 (coordinates) and 5 (pressure).  Velocities and the flag column are
    not compared.
  * coordinates must agree line for line and the step counts must be equal;
  * per value: diff = ref - sim, percent = 100 * diff / sim; the entry with the
    largest |percent| is reported (a NaN wins);
  * a file fails when that |percent| exceeds the tolerance (default 1 %) or is
    not finite.  Exit status 0 = both pass, 1 = anything else.

The reference pressure may also be given as one of this repo's compact
fixtures (`*.pressure.f64.npz`, see tests/golden/make_golden.py) for the two
decks whose golden final_state.dat is missing from the reference checkout.

Usage (same flags as the reference tool):
  check_results.py --ref-av-vels-file R1 --ref-final-state-file R2 \
                   --av-vels-file S1 --final-state-file S2 [--tolerance PCT]
"""
import argparse
import sys

import numpy as np


def load_av_vels(path):
    return np.atleast_1d(np.loadtxt(path, usecols=[1]))


def load_final_state(path):
    """-> (coords or None, pressure).  coords is (n, 2)."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return None, np.asarray(z["pressure"], dtype=np.float64).ravel()
    table = np.atleast_2d(np.loadtxt(path, usecols=[0, 1, 5]))
    return table[:, :2], table[:, 2]


def worst_deviation(ref, sim):
    """Largest relative deviation in the reference tool's convention."""
    with np.errstate(divide="ignore", invalid="ignore"):
        delta = ref - sim
        percent = 100.0 * (delta / (ref - delta))
    k = int(np.argmax(np.abs(percent)))
    return {"index": k, "delta": float(delta[k]), "percent": float(percent[k]),
            "sim": float(sim[k]), "ref": float(ref[k]), "total": float(np.sum(np.abs(delta)))}


def passes(dev, tolerance):
    return bool(np.isfinite(dev["percent"]) and abs(dev["percent"]) <= tolerance)


def compare(ref_av, ref_fs, sim_av, sim_fs, tolerance=1.0, out=sys.stdout):
    """Returns (ok, av_dev, fs_dev); prints a report in the reference tool's wording."""
    av_ref = load_av_vels(ref_av)
    av_sim = load_av_vels(sim_av)
    xy_ref, p_ref = load_final_state(ref_fs)
    xy_sim, p_sim = load_final_state(sim_fs)

    if xy_ref is not None and xy_sim is not None:
        if xy_ref.shape != xy_sim.shape or np.any(xy_ref != xy_sim):
            print("Final state files coordinates were not the same", file=out)
            return False, None, None
    elif p_ref.size != p_sim.size:
        print("Final state files coordinates were not the same", file=out)
        return False, None, None
    if av_ref.size != av_sim.size:
        print("Different number of steps in av_vels files", file=out)
        return False, None, None

    a = worst_deviation(av_ref, av_sim)
    print("Total difference in av_vels : %.12E" % a["total"], file=out)
    print("Biggest difference (at step %d) : %.12E" % (a["index"], a["delta"]), file=out)
    print("  %.12E vs. %.12E = %.2g%%" % (a["sim"], a["ref"], a["percent"]), file=out)
    print(file=out)

    f = worst_deviation(p_ref, p_sim)
    if xy_sim is not None:
        cx, cy = int(xy_sim[f["index"], 0]), int(xy_sim[f["index"], 1])
    else:
        cx, cy = -1, -1
    f["coord"] = (cx, cy)
    print("Total difference in final_state : %.12E" % f["total"], file=out)
    print("Biggest difference (at coord (%d,%d)) : %.12E" % (cx, cy, f["delta"]), file=out)
    print("  %.12E vs. %.12E = %.2g%%" % (f["sim"], f["ref"], f["percent"]), file=out)
    print(file=out)

    fs_ok, av_ok = passes(f, tolerance), passes(a, tolerance)
    if not fs_ok:
        print("final state failed check", file=out)
    if not av_ok:
        print("av_vels failed check", file=out)
    if fs_ok and av_ok:
        print("Both tests passed!", file=out)
    return fs_ok and av_ok, a, f


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0], fromfile_prefix_chars="@")
    ap.add_argument("--tolerance", type=float, default=1.0, help="percent")
    ap.add_argument("--ref-av-vels-file", required=True)
    ap.add_argument("--ref-final-state-file", required=True)
    ap.add_argument("--av-vels-file", required=True)
    ap.add_argument("--final-state-file", required=True)
    ns = ap.parse_args(argv)
    ok, _, _ = compare(ns.ref_av_vels_file, ns.ref_final_state_file,
                       ns.av_vels_file, ns.final_state_file, ns.tolerance)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
