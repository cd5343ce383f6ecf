#!/usr/bin/env python3
"""The launches the round's counter passes are taken on (rocprofv3 --pmc ... -- python3 tools/profile_run.py):
  1024x1024 deck, default engine (lbm_regtile: one launch per run): runs of 1, 18, 400 and 2000 steps -- the straight line
      through their counters gives "per launch" and "per step";
  8192x8192 synthetic deck, default kernel: 4 runs of 32 steps (= 16 launches of 8 steps);
  8192x8192, lbm_march: 1 run of 32 steps (8 launches of 4), for comparison.
tools/collect_counters.py reads the passes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import advanced_hpc_lbm_amd as L  # noqa: E402
from make_deck import obstacle_map  # noqa: E402

REGTILE_STEPS = (1, 18, 400, 2000)

if __name__ == "__main__":
    p = L.read_params(os.path.join(ROOT, "input_1024x1024.params"))
    ob = L.read_obstacles(os.path.join(ROOT, "obstacles_1024x1024.dat"), p)
    with L.Lattice(p, ob) as lat:
        for n in REGTILE_STEPS:
            lat.run(n)
        assert int(lat.info("engine_last")) == 3
    n = 8192
    p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85)
    ob = obstacle_map(n, n)
    with L.Lattice(p, ob) as lat:
        for _ in range(4):
            lat.run(32)
        print("8192x8192 default: steps per pass", int(lat.info("time_block_active")), "lbm_wave" if lat.info("march_kernel") else "lbm_march",
              "columns per lane", int(lat.info("wave_cols_active")), "rows per chunk", int(lat.info("wave_rows")))
    with L.Lattice(p, ob) as lat:
        lat.set_option("march_kernel", 0)
        lat.set_option("time_block", 4)
        lat.run(32)
