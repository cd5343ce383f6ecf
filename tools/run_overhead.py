#!/usr/bin/env python3
"""Fixed cost of one lbm_run on a shipped deck: wall clock around the Python call, wall clock inside
the library, and the HIP-event time on the library's stream, for runs of n steps (median of `reps`).
A straight-line fit over n gives the per-step time and what a run costs before its first step.
    python tools/run_overhead.py [deck=1024x1024] [reps=30] [n ...]"""
import os
import statistics
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import advanced_hpc_lbm_amd as L  # noqa: E402
try:
    import torch
except ImportError:
    torch = None

deck = sys.argv[1] if len(sys.argv) > 1 else "1024x1024"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
ns = [int(v) for v in sys.argv[3:]] or [1, 2, 20, 100]
p = L.read_params(os.path.join(ROOT, f"input_{deck}.params"))
ob = L.read_obstacles(os.path.join(ROOT, f"obstacles_{deck}.dat"), p)
rows = []
with L.Lattice(p, ob) as lat:
    lat.run(20)
    for n in ns:
        py, lib, gpu, fen = [], [], [], []
        for _ in range(reps):
            if torch is not None:
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            lat.run(n)
            t1 = time.perf_counter()
            py.append((t1 - t0) * 1e6)
            if torch is not None:                # what bench.py's closing torch.cuda.synchronize() adds
                torch.cuda.synchronize()
                fen.append((time.perf_counter() - t1) * 1e6)
            g, w = lat.last_run_ms()
            lib.append(w * 1e3)
            gpu.append(g * 1e3)
        rows.append((n, statistics.median(py), statistics.median(lib), statistics.median(gpu)))
        print(f"{deck} n={n:5d}: python {rows[-1][1]:8.1f} us   library {rows[-1][2]:8.1f} us   gpu events {rows[-1][3]:8.1f} us"
              f"   torch sync after {statistics.median(fen) if fen else 0:5.1f} us"
              f"   [{['', 'streaming', 'lbm_resident', 'lbm_regtile'][int(lat.info('engine_last'))]}]", flush=True)
if len(rows) >= 2:
    (n0, p0, l0, g0), (n1, p1, l1, g1) = rows[0], rows[-1]
    for name, a, b in (("python", p0, p1), ("library", l0, l1), ("gpu events", g0, g1)):
        step = (b - a) / (n1 - n0)
        print(f"{name:10s}: {step:7.3f} us per step, {a - step * n0:7.1f} us per run")
# the driver's sequence (python bench.py --steps 20 --warmup 5): a fresh lattice, ONE warm-up run, ONE timed run
for trial in range(3):
    with L.Lattice(p, ob) as lat:
        lat.run(5)
        if torch is not None:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        lat.run(20)
        if torch is not None:
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e6
        g, w = lat.last_run_ms()
        t0 = time.perf_counter()
        lat.run(20)
        if torch is not None:
            torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) * 1e6
        g2, w2 = lat.last_run_ms()
        print(f"fresh lattice, run(5) then run(20): python+sync {dt:7.1f} us  library {w * 1e3:7.1f}  gpu events {g * 1e3:7.1f}"
              f"   | the same again: {dt2:7.1f}  {w2 * 1e3:7.1f}  {g2 * 1e3:7.1f}", flush=True)
