#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, as
MI355X_MICROARCH.md §HBM prescribes) of ONE workload, merged into profiles/hbm_traffic.json.

  summarize_pmc.py <tag> <NXxNY> <fetch_dir> <write_dir>

FETCH_SIZE is doubled (gfx950 reports half the bytes of 16-B-per-lane streaming reads, LDS-DMA alike),
both x 1024 (KiB).  Kernels: lbm_march (4 steps per launch), lbm_sweep2 (2), lbm_sweep (1); the most
frequent grid size of each is taken as the whole-lattice launch."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, wl, fdir, wdir = sys.argv[1:5]
nx, ny = (int(v) for v in wl.split("x"))
cells = nx * ny
STEPS = {"lbm_march": 4, "lbm_sweep2": 2, "lbm_sweep": 1, "lbm_wave4": 4, "lbm_wave6": 6, "lbm_wave8": 8}
vals = collections.defaultdict(dict)
for which, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != which:
            continue
        full = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lbm::", "")
        kern = full.split("<")[0]
        if kern == "lbm_wave":
            kern += full.split("<")[1].split(",")[0].strip()
        if kern in STEPS:
            agg[(kern, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    for kern in STEPS:
        grids = {g: v for (k, g), v in agg.items() if k == kern}
        if grids:
            g, v = max(grids.items(), key=lambda kv: len(kv[1]))
            v.sort()
            vals[kern][which] = v[len(v) // 2]
            vals[kern]["grid_threads"] = g
            vals[kern]["launches_" + which] = len(v)
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
traffic = json.load(open(path)) if os.path.exists(path) else {}
for kern, c in vals.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    steps = STEPS[kern]
    fetch_b, write_b = 2.0 * c["FETCH_SIZE"] * 1024.0, c["WRITE_SIZE"] * 1024.0
    traffic.setdefault(wl, {})[kern] = {
        "grid_threads": c["grid_threads"], "steps_per_launch": steps,
        "FETCH_SIZE_KiB_raw": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
        "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b,
        "hbm_bytes_per_launch": fetch_b + write_b,
        "hbm_bytes_per_lattice_update": (fetch_b + write_b) / (cells * steps),
        "equiv_72B_bytes_per_launch": 72.0 * cells * steps, "round": tag,
        "collected": os.environ.get("PMC_NOTE", "")}
json.dump(traffic, open(path, "w"), indent=1)
print(json.dumps({k: v for k, v in traffic.get(wl, {}).items()}, indent=1))
