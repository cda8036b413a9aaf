#!/usr/bin/env python3
"""Pairs the LAUNCH lines of tools/coherence_probe.py with the walk kernels of a rocprofv3 kernel trace (in start order).
usage: tools/coherence_report.py <probe stdout> <kernel_trace.csv>"""
import csv
import sys

launches = [l.split()[1:] for l in open(sys.argv[1]) if l.startswith("LAUNCH ")]
rows = []
with open(sys.argv[2]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].split()[-1]
        if name.startswith(("k_closest", "k_any")):
            rows.append((int(r["Start_Timestamp"]), name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
assert len(rows) == len(launches), (len(rows), len(launches))
base = {}
for (_, kname, ms), (kind, bounce, order, n) in zip(rows, launches):
    key = (kind, bounce)
    base.setdefault(key, ms)
    print(f"{kind:8s} {bounce:8s} {order:7s} {kname:10s} {int(n):9d} rays {ms:8.3f} ms  {int(n) / ms / 1e6:7.3f} Grays/s  x{base[key] / ms:5.2f} vs path order")
