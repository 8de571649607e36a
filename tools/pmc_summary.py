#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

paths = sum((glob.glob(p + "/**/*counter_collection.csv", recursive=True) for p in sys.argv[1:]), [])
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
seen = set()
for path in paths:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"][:70]
        c = acc[k][r["Counter_Name"]]
        c[0] += float(r["Counter_Value"])
        c[1] += 1
        if (path, r["Dispatch_Id"]) not in seen:
            seen.add((path, r["Dispatch_Id"]))
            dur[k][0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            dur[k][1] += 1
for k, cs in sorted(acc.items(), key=lambda kv: -dur[kv[0]][0]):
    if "adamvs" not in k:
        continue
    print("%s   avg %.1f us x %d" % (k, dur[k][0] / dur[k][1], dur[k][1]))
    print("   " + "  ".join("%s=%.4g" % (n, v[0] / v[1]) for n, v in sorted(cs.items())))
