#!/bin/bash
set -o pipefail
O=gpurun_out/r02r
mkdir -p $O
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 400 python - > $O/small.log 2>&1 <<PY
import os, sys
sys.path.insert(0, os.getcwd())
import advanced_hpc_lbm_amd as L
for deck, tiles in (("128x128", (11, 21, 22, 42, 44, 84, 164)), ("128x256", (11, 21, 22, 42, 44, 84)), ("256x256", (41, 42, 44, 82, 84, 164, 324)), ("1024x1024", (644, 324))):
    p = L.read_params(f"input_{deck}.params"); ob = L.read_obstacles(f"obstacles_{deck}.dat", p)
    with L.Lattice(p, ob) as lat:
        lat.run(4000)
        best = min(lat.run(4000) is None or lat.last_run_ms()[0] for _ in range(3))
        print(deck, "streaming %.3f us/step" % (best * 1e3 / 4000), flush=True)
    for tile in tiles:
        with L.Lattice(p, ob) as lat:
            try:
                lat.set_option("regtile", tile)
            except L.LbmError as e:
                print(deck, "tile", tile, "no"); continue
            lat.set_option("engine", 3)
            lat.run(4000)
            best = min(lat.run(4000) is None or lat.last_run_ms()[0] for _ in range(3))
            print(deck, "regtile", tile, "%.3f us/step  %.1f GLUPS" % (best * 1e3 / 4000, p.nx*p.ny*4000/best/1e6), flush=True)
PY
cat $O/small.log
