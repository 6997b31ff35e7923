#!/usr/bin/env python3
"""Per-call durations of selected kernels from a rocprofv3 kernel trace (ms), in launch order."""
import csv, glob, sys, collections
d, pats = sys.argv[1], sys.argv[2:]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = collections.defaultdict(list)
for r in rows:
    for p in pats:
        if p in r["Kernel_Name"]:
            out[p].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for p in pats:
    print(p, " ".join(f"{x:.2f}" for x in out[p]))
