#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv -> per kernel: launches, mean duration, mean gap to the NEXT kernel's start (the dependent-launch
boundary) over the last `frac` of the trace (the timed replays), plus the totals.

    python tools/trace_gaps.py <kernel_trace.csv> [tail_fraction=0.5]
"""
import csv
import re
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for i, (s, e, k) in enumerate(rows):
    name = re.sub(r"\(.*", "", k.replace("adamvs::", "").replace("void ", ""))[:90]
    dur[name] += (e - s) / 1e3
    cnt[name] += 1
    if i + 1 < len(rows):
        gap[name] += max(rows[i + 1][0] - e, 0) / 1e3 if rows[i + 1][0] - e < 200000 else 0.0
tot_d, tot_g = sum(dur.values()), sum(gap.values())
span = (rows[-1][1] - rows[0][0]) / 1e3
print("span %.1f us, kernel time %.1f us (%.1f %%), gaps %.1f us (%.1f %%), %d launches" % (span, tot_d, 100 * tot_d / span, tot_g, 100 * tot_g / span, len(rows)))
for name in sorted(dur, key=lambda n: -(dur[n] + gap[n])):
    print("%7d x  dur %8.2f us  gap %6.2f us  total %9.1f us (%5.2f %%)  %s" % (cnt[name], dur[name] / cnt[name], gap[name] / cnt[name],
                                                                               dur[name] + gap[name], 100 * (dur[name] + gap[name]) / span, name))
