#!/usr/bin/env python3
"""us/step of lbm_regtile on a shipped deck for a list of tilings (rows per tile x 10 + rows per wave; none = the
default), best of three runs by HIP events.  LBM_RESIDENT_DEBUG=1/2/3 selects the timing-only variants (wrong
results), LBM_REGTILE_STATS=1 prints how many polls found their mail missing.
    python tools/regtile_timing.py deck steps [tiling ...]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import advanced_hpc_lbm_amd as L  # noqa: E402

deck, steps = sys.argv[1], int(sys.argv[2])
tilings = [int(v) for v in sys.argv[3:]] or [None]
p = L.read_params(os.path.join(ROOT, f"input_{deck}.params"))
ob = L.read_obstacles(os.path.join(ROOT, f"obstacles_{deck}.dat"), p)
for t in tilings:
    with L.Lattice(p, ob) as lat:
        lat.set_option("engine", 3)
        if t is not None:
            try:
                lat.set_option("regtile", t)
            except L.LbmError as e:
                print(deck, t, "not usable:", e)
                continue
        lat.run(steps)
        best = 1e9
        for _ in range(3):
            lat.run(steps)
            g, w = lat.last_run_ms()
            best = min(best, g)
        assert int(lat.info("engine_last")) == 3
        print(f"dbg={os.environ.get('LBM_RESIDENT_DEBUG', '0')} async={int(lat.info('regtile_async'))} {deck} regtile {int(lat.info('regtile'))} "
              f"{best * 1e3 / steps:.3f} us/step  {p.nx * p.ny * steps / best / 1e6:.1f} GLUPS", flush=True)
