#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv -> one short line per kernel: python tools/show_kernel_stats.py file.csv [regex]"""
import csv
import re
import sys

pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for r in csv.reader(open(sys.argv[1])):
    if r[0] == "Name" or (pat and not pat.search(r[0])):
        continue
    name = re.sub(r"\(adamvs::SlotArgs.*", "", r[0].replace("adamvs::", "").replace("void ", ""))
    print("%9.1f us x %5s  (%5.2f %%)  %s" % (float(r[3]) / 1e3, r[1], float(r[4]), name[:150]))
