#!/usr/bin/env python3
"""Where a frame's time goes between its kernels, from a rocprofv3 --kernel-trace CSV: the frames are cut at the projection
kernel, and for the last <frames> of them the kernel time, the gaps between consecutive kernels (end -> next start) and the gap
after each kernel (by kernel name) are averaged.  usage: kernel_gaps.py <dir> <frames> [project-kernel substring] [timeline [cuts]]"""
import collections
import csv
import glob
import sys

d, frames = sys.argv[1], int(sys.argv[2])
cut = sys.argv[3] if len(sys.argv) > 3 else "k_project"
f = (glob.glob(d + "/*/*_kernel_trace.csv") + glob.glob(d + "/*_kernel_trace.csv"))[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if cut in r[2]]
starts = starts[-(frames + 1):]
busy = gap = 0.0
after = collections.defaultdict(list)
n_k = 0
for a, b in zip(starts[:-1], starts[1:]):
    for i in range(a, b):
        s, e, name = rows[i]
        busy += (e - s) / 1e3
        g = (rows[i + 1][0] - e) / 1e3
        gap += g
        after[name].append(g)
        n_k += 1
nf = len(starts) - 1
span = (rows[starts[-1]][0] - rows[starts[0]][0]) / 1e3 / nf
print(f"{nf} frames: {span:.1f} us per frame = {busy / nf:.1f} us in kernels + {gap / nf:.1f} us between them; {n_k / nf:.1f} kernels per frame, {gap / max(n_k, 1):.2f} us per boundary")
for name, v in sorted(after.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  after {name:46s} {sum(v) / nf:7.1f} us per frame  ({len(v) / nf:4.1f} x {sum(v) / len(v):6.2f} us)")
if len(sys.argv) > 4 and sys.argv[4] == "timeline":   # the last whole frame, kernel by kernel: start offset, duration, gap to the next
    a, b = starts[-1 - (int(sys.argv[5]) if len(sys.argv) > 5 else 1)], starts[-1]
    t0 = rows[a][0]
    for i in range(a, b):
        s, e, name = rows[i]
        print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  gap {(rows[i + 1][0] - e) / 1e3:6.1f}  {name}")
