#!/usr/bin/env python3
"""Collapses the counter_collection CSVs of profiles/run_pmc.sh into one JSON: per kernel, the
mean of every counter per dispatch (kernels of the repo only: names starting with k_ or void k_).
FETCH_SIZE / WRITE_SIZE are reported in bytes with the gfx950 corrections of
MI355X_MICROARCH.md (values are KB; FETCH_SIZE counts half the bytes of wide coalesced reads —
both raw and doubled figures are given)."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in glob.glob(f"{out}/*/**/*counter_collection.csv", recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        m = re.match(r"(?:void )?(k_[a-z_0-9]+(?:<[^>]*>)?)\(", name)
        if not m:
            continue
        short = m.group(1)
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
        key = (f, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[short].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
res = {}
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    d["dispatches_seen"] = max(len(v) for v in cs.values())
    d["mean_duration_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes_raw"] = d["FETCH_SIZE"] * 1024
        d["hbm_read_bytes_x2_wide_loads"] = d["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024
    if "GRBM_GUI_ACTIVE" in d:
        d["clock_GHz_est"] = d["GRBM_GUI_ACTIVE"] / 8 / (d["mean_duration_us_under_pmc"] * 1e3)
    res[k] = d
json.dump(res, sys.stdout, indent=1, sort_keys=True)
print()
