#!/usr/bin/env python3
"""ONE lbm_wave configuration, a few launches, for rocprofv3 counter passes: python3 tools/wave_one.py N K COLS ROWS [LAUNCHES]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import advanced_hpc_lbm_amd as L  # noqa: E402
from make_deck import obstacle_map  # noqa: E402
n, k, cols, rows = (int(v) for v in sys.argv[1:5])
launches = int(sys.argv[5]) if len(sys.argv) > 5 else 6
p = L.Param(n, n, 1000, 10, 0.1, 0.01, 1.85)
with L.Lattice(p, obstacle_map(n, n)) as lat:
    lat.set_option("march_kernel", 1)
    lat.set_option("time_block", k)
    lat.set_option("wave_cols", cols)
    lat.set_option("wave_rows", rows)
    lat.run(k * launches)
    g = lat.last_run_ms()[0]
    print(f"{n}x{n} lbm_wave<{k}> x{int(lat.info('wave_cols_active'))} rows {rows}: {g * 1e3 / (k * launches):.2f} us/step "
          f"{n * n * k * launches / g / 1e6:.1f} GLUPS")
