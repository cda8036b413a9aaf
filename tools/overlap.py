#!/usr/bin/env python3
"""Concurrency picture from a rocprofv3 --kernel-trace CSV: wall time, time with 0/1/2/.. kernels resident, and per
kernel the summed duration.  usage: tools/overlap.py <kernel_trace.csv> [skip_fraction]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.34
ev = []
t_all0 = min(int(r["Start_Timestamp"]) for r in rows)
t_all1 = max(int(r["End_Timestamp"]) for r in rows)
t_cut = t_all0 + (t_all1 - t_all0) * skip  # drop warm-up
dur = defaultdict(float)
for r in rows:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if a < t_cut:
        continue
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur[n] += (b - a) * 1e-6
    kind = "trav" if n in ("k_closest", "k_shadow") else "other"
    ev.append((a, 1, kind))
    ev.append((b, -1, kind))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
state = {"trav": 0, "other": 0}
hist = defaultdict(float)
prev = t0
for t, d, kind in ev:
    hist[(min(state["trav"], 3), min(state["other"], 3))] += (t - prev) * 1e-6
    state[kind] += d
    prev = t
wall = (t1 - t0) * 1e-6
print(f"wall {wall:.1f} ms")
for k in sorted(hist):
    print(f"  traversal kernels {k[0]}{'+' if k[0]==3 else ' '} other kernels {k[1]}{'+' if k[1]==3 else ' '}: {hist[k]:8.2f} ms  {100*hist[k]/wall:5.1f} %")
for n, v in sorted(dur.items(), key=lambda x: -x[1]):
    print(f"  {n:24s} {v:9.2f} ms summed")
