#!/usr/bin/env python3
"""Per-frame kernel time breakdown from a rocprofv3 --kernel-trace CSV.  usage: kernel_breakdown.py <dir> <frames>"""
import collections
import csv
import glob
import sys

d, frames = sys.argv[1], float(sys.argv[2])
f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
acc = collections.defaultdict(list)
t0, t1 = None, None
for r in csv.DictReader(open(f)):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    acc[r["Kernel_Name"].split("(")[0][-40:]].append((e - s) / 1e3)
rows = [(sum(v) / frames, k, len(v) / frames, sum(v) / len(v)) for k, v in acc.items()]
for r in sorted(rows, reverse=True)[: int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    print("per-frame %7.1f us  %-42s calls/frame=%5.1f avg=%7.1f us" % r)
print("sum per frame us %.1f" % sum(r[0] for r in rows))
