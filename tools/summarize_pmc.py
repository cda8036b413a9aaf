#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and prints derived figures.
usage: tools/summarize_pmc.py <dir with one sub-directory per pass> [out.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(set))
dur = defaultdict(float)
for path in glob.glob(os.path.join(root, "*", "*", "*_counter_collection.csv")):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]].add(row["Dispatch_Id"])
for path in glob.glob(os.path.join(root, "sq1", "*", "*_kernel_trace.csv")):
    for row in csv.DictReader(open(path)):
        dur[row["Kernel_Name"].split("(")[0].replace("void ", "")] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-6
out = {}
for k, c in sorted(tot.items()):
    if not k.startswith("k_"):
        continue
    d = dict(c)
    d["ms_profiled_sq1"] = dur.get(k, 0.0)
    d["dispatches"] = max(len(v) for v in calls[k].values())
    if c.get("SQ_ACTIVE_INST_VALU"):
        # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = mean active lanes per VALU instruction cycle
        d["valu_active_lanes_of_64"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"] if c.get("SQ_THREAD_CYCLES_VALU") else None
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_WAVE_CYCLES"):
        d["mean_waves_per_busy_cycle"] = c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"]
    if c.get("SQ_WAVE_CYCLES"):
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c.get(n):
                d[n + "_frac_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    if c.get("TCC_HIT_sum") is not None and (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) > 0:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "FETCH_SIZE" in c:
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE (KiB) reads 1/2 of a wide coalesced stream's bytes on gfx950 -> doubled
        d["hbm_read_bytes_corrected"] = 2 * c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c:
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    out[k] = d
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
