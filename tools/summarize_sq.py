#!/usr/bin/env python3
"""Counters of one rocprofv3 --pmc pass for the LONGEST dispatch of each kernel (the timed launch of a bench run),
merged into profiles/r02_sq_counters.json under a label.
    summarize_sq.py <label> <pmc_dir> [kernel-name-substring]"""
import collections
import csv
import glob
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
label, d = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else "lbm_"
f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
disp = collections.defaultdict(lambda: {"us": 0.0, "counters": {}})
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if want not in k:
        continue
    e = disp[(k, r["Dispatch_Id"])]
    e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    e["counters"][r["Counter_Name"]] = e["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
best = {}
for (k, _), e in disp.items():
    if k not in best or e["us"] > best[k]["us"]:
        best[k] = e
out = {k: {"launch_us": round(e["us"], 1), **{c: v for c, v in sorted(e["counters"].items())}} for k, e in best.items()}
p = os.path.join(ROOT, "profiles", "r02_sq_counters.json")
j = json.load(open(p)) if os.path.exists(p) else {}
j[label] = out
json.dump(j, open(p, "w"), indent=1)
for k, cs in out.items():
    print(k, cs)
