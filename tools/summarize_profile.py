#!/usr/bin/env python3
"""Condenses rocprofv3 output (csv) into the small summaries committed under profiles/.

  summarize_profile.py <round-tag> <kernel_trace_dir> [<pmc_fetch_dir> <pmc_write_dir>]

Writes profiles/<tag>_kernel_stats.csv (rocprofv3's own --stats table, copied),
profiles/<tag>_sweep_by_grid.json (mean/median launch time of the sweep kernel per grid size,
i.e. per workload) and, with the two PMC passes, profiles/hbm_traffic.json: HBM bytes per
launch = 2 x FETCH_SIZE KiB (gfx950 reports half the bytes of 16-B-per-lane streaming reads,
MI355X_MICROARCH.md §HBM) + WRITE_SIZE KiB, both x 1024.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, trace_dir = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

stats = glob.glob(os.path.join(trace_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
trace = glob.glob(os.path.join(trace_dir, "**", "*_kernel_trace.csv"), recursive=True)[0]
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    if "lbm_sweep" in r["Kernel_Name"] or "lbm_sweep2" in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
by_grid = {}
for (k, g), v in sorted(dur.items()):
    v.sort()
    by_grid[f"{k} grid={g}"] = {"launches": len(v), "mean_ns": sum(v) / len(v), "median_ns": v[len(v) // 2],
                                "min_ns": v[0], "max_ns": v[-1]}
json.dump(by_grid, open(os.path.join(out, f"{tag}_sweep_by_grid.json"), "w"), indent=1)
print(json.dumps(by_grid, indent=1))

if len(sys.argv) >= 5:
    vals = {}
    for which, d in (("FETCH_SIZE", sys.argv[3]), ("WRITE_SIZE", sys.argv[4])):
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "lbm_sweep" in r["Kernel_Name"] and r["Counter_Name"] == which:
                kern = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lbm::", "").split("<")[0]
                agg[(kern, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
        for key, v in agg.items():
            v.sort()
            vals.setdefault(key, {})[which] = v[len(v) // 2]
    # (kernel, grid threads) -> workload: lbm_sweep covers cells/V cells per thread (V in 1,2,4),
    # lbm_sweep2 covers 4 cells per thread and TWO steps per launch
    path = os.path.join(out, "hbm_traffic.json")
    traffic = json.load(open(path)) if os.path.exists(path) else {}
    for (kern, g), c in sorted(vals.items()):
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        fetch_b = 2.0 * c["FETCH_SIZE"] * 1024.0
        write_b = c["WRITE_SIZE"] * 1024.0
        for name, cells in (("1024x1024", 1024 * 1024), ("8192x8192", 8192 * 8192)):
            if cells % g == 0 and cells // g in (1, 2, 4):
                steps = 2 if kern == "lbm_sweep2" else 1
                traffic.setdefault(name, {})[kern] = {
                    "grid_threads": g, "steps_per_launch": steps,
                    "FETCH_SIZE_KiB_raw": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"],
                    "fetch_bytes_corrected_x2": fetch_b, "write_bytes": write_b,
                    "hbm_bytes_per_launch": fetch_b + write_b,
                    "algorithmic_bytes_per_launch": 72.0 * cells * steps,
                    "hbm_bytes_per_lattice_update": (fetch_b + write_b) / (cells * steps), "round": tag}
    json.dump(traffic, open(path, "w"), indent=1)
    print(json.dumps(traffic, indent=1))
