#!/usr/bin/env python3
"""rocprofv3 --pmc passes of tools/profile_run.py -> profiles/<tag>_sq_counters.json (everything), and the two small files
bench.py reads: profiles/hbm_traffic.json (FETCH_SIZE doubled per MI355X_MICROARCH.md, HBM section; + WRITE_SIZE; KiB x 1024)
and profiles/kernel_counters.json (SQ_INSTS_VALU).

    collect_counters.py <tag> <note> <pass_dir> [<pass_dir> ...]      (one directory per rocprofv3 run)"""
import collections
import csv
import glob
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from profile_run import REGTILE_STEPS  # noqa: E402

tag, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]


def kernel_key(name):
    full = name.split("(")[0].replace("void ", "").replace("lbm::", "")
    base = full.split("<")[0]
    if base == "lbm_wave":
        args = [a.strip() for a in full.split("<")[1].rstrip(">").split(",")]
        return "lbm_wave" + args[0] + ("x2" if len(args) > 3 and args[3] == "2" else "")
    return base


# per pass: kernel -> list of dispatches (in dispatch order) -> {counter: value}, duration
allc = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> dispatch index -> {counter: v, "us": t}
for d in dirs:
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        k = kernel_key(r["Kernel_Name"])
        if not (k.startswith("lbm_wave") or k in ("lbm_regtile", "lbm_march", "lbm_sweep2", "lbm_sweep")):
            continue
        e = per[k][int(r["Dispatch_Id"])]
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        e["grid"] = int(r["Grid_Size"])
    for k, disp in per.items():
        for i, did in enumerate(sorted(disp)):
            allc[k][i].update({c: v for c, v in disp[did].items() if c not in ("us", "grid")})
            allc[k][i].setdefault("us_by_pass", []).append(disp[did]["us"])
            allc[k][i]["grid"] = disp[did]["grid"]

out = {"note": f"rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/profile_run.py, one pass per counter group ({note}); "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"}
traffic_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
counters_path = os.path.join(ROOT, "profiles", "kernel_counters.json")
traffic = json.load(open(traffic_path)) if os.path.exists(traffic_path) else {}
kcount = json.load(open(counters_path)) if os.path.exists(counters_path) else {}


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


for k, disp in sorted(allc.items()):
    if k == "lbm_regtile":
        # one launch per run of 1, 18, 400, 2000 steps: straight line through the first and the last but one / last
        rows = {}
        for i, n in enumerate(REGTILE_STEPS):
            if i in disp:
                rows[n] = {c: v for c, v in disp[i].items() if c not in ("us_by_pass",)}
                rows[n]["launch_us"] = median(disp[i]["us_by_pass"])
        out["lbm_regtile, 1024x1024 deck, one launch per run of n steps"] = rows
        if 18 in rows and 2000 in rows:
            def line(c):
                per_step = (rows[2000][c] - rows[18][c]) / (2000 - 18)
                return rows[18][c] - 18 * per_step, per_step
            ent = traffic.setdefault("1024x1024", {}).setdefault("lbm_regtile", {})
            if "FETCH_SIZE" in rows[18] and "WRITE_SIZE" in rows[18]:
                f0, f1 = line("FETCH_SIZE")
                w0, w1 = line("WRITE_SIZE")
                ent.update({"note": "ONE launch per lbm_run: bytes = fixed (the lattice loaded and stored once) + steps x per-step (the tiles' mail: "
                                    "16-byte sc1 granules through memory); straight line through the launches of 18 and 2000 steps of one PMC "
                                    "pass each (FETCH_SIZE doubled per the guide, WRITE_SIZE as is, KiB x 1024)",
                            "fetch_fixed_bytes_corrected_x2": 2 * f0 * 1024, "fetch_bytes_per_step_corrected_x2": 2 * f1 * 1024,
                            "write_fixed_bytes": w0 * 1024, "write_bytes_per_step": w1 * 1024,
                            "hbm_fixed_bytes_per_launch": (2 * f0 + w0) * 1024, "hbm_bytes_per_step": (2 * f1 + w1) * 1024,
                            "hbm_bytes_per_lattice_update_in_the_loop": (2 * f1 + w1) * 1024 / (1024 * 1024),
                            "round": tag, "collected": note})
            if "SQ_INSTS_VALU" in rows[18]:
                v0, v1 = line("SQ_INSTS_VALU")
                kcount.setdefault("1024x1024", {})["lbm_regtile"] = {
                    "valu_wave_insts_per_step": v1, "valu_wave_insts_fixed_per_launch": v0, "waves": rows[2000].get("SQ_WAVES"),
                    "source": f"profiles/{tag}_sq_counters.json: launches of 18 and 2000 steps", "round": tag}
        continue
    # streaming kernels: the dispatches of the 8192x8192 runs (the largest grid), median over dispatches
    big = max(e["grid"] for e in disp.values())
    sel = [e for e in disp.values() if e["grid"] == big]
    row = {"launches": len(sel), "grid_threads": big, "launch_us": median([median(e["us_by_pass"]) for e in sel])}
    for c in sorted({c for e in sel for c in e if c not in ("us_by_pass", "grid")}):
        row[c] = median([e[c] for e in sel if c in e])
    out[f"{k}, 8192x8192"] = row
    steps = {"lbm_march": 4, "lbm_sweep2": 2, "lbm_sweep": 1}.get(k, int(k[8]) if k.startswith("lbm_wave") else 1)
    cells = 8192 * 8192
    if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
        fb, wb = 2.0 * row["FETCH_SIZE"] * 1024.0, row["WRITE_SIZE"] * 1024.0
        traffic.setdefault("8192x8192", {})[k] = {
            "grid_threads": big, "steps_per_launch": steps, "FETCH_SIZE_KiB_raw": row["FETCH_SIZE"], "WRITE_SIZE_KiB": row["WRITE_SIZE"],
            "fetch_bytes_corrected_x2": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb,
            "hbm_bytes_per_lattice_update": (fb + wb) / (cells * steps), "equiv_72B_bytes_per_launch": 72.0 * cells * steps,
            "round": tag, "collected": note}
    if "SQ_INSTS_VALU" in row:
        kcount.setdefault("8192x8192", {})[k] = {"valu_wave_insts_per_launch": row["SQ_INSTS_VALU"], "waves": row.get("SQ_WAVES"),
                                                 "valu_lane_insts_per_lattice_update": row["SQ_INSTS_VALU"] * 64 / (cells * steps),
                                                 "source": f"profiles/{tag}_sq_counters.json", "round": tag}

json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_sq_counters.json"), "w"), indent=1)
json.dump(traffic, open(traffic_path, "w"), indent=1)
json.dump(kcount, open(counters_path, "w"), indent=1)
for k, v in out.items():
    if k != "note":
        print(k, json.dumps(v)[:600])
