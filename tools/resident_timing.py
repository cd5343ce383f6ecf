#!/usr/bin/env python3
"""us/step of the resident kernel on the shipped decks for a list of tilings (and, with
LBM_RESIDENT_DEBUG set, of its no-wait / no-send timing variants).  python tools/resident_timing.py deck steps tile..."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import advanced_hpc_lbm_amd as L  # noqa: E402

deck, steps = sys.argv[1], int(sys.argv[2])
tiles = [tuple(int(v) for v in t.split("x")) for t in sys.argv[3:]] or [None]
p = L.read_params(os.path.join(ROOT, f"input_{deck}.params"))
ob = L.read_obstacles(os.path.join(ROOT, f"obstacles_{deck}.dat"), p)
for tile in tiles:
    with L.Lattice(p, ob) as lat:
        lat.set_option("engine", 2)
        if tile:
            try:
                lat.set_option("resident_tile", tile[0] * 100000 + tile[1] * 10 + tile[2])
            except L.LbmError as e:
                print(deck, tile, "not usable:", e)
                continue
        lat.run(steps)
        best = 1e9
        for _ in range(3):
            lat.run(steps)
            g, w = lat.last_run_ms()
            best = min(best, g)
        print(f"{deck} tile {int(lat.info('resident_tile'))} dbg={os.environ.get('LBM_RESIDENT_DEBUG','0')}: {best * 1e3 / steps:.3f} us/step gpu "
              f"({p.nx * p.ny * steps / best / 1e6:.1f} GLUPS)", flush=True)
