#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel time, busy time of the union of all
kernel intervals, and how much of it had two or more kernels in flight (stream overlap)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per = defaultdict(lambda: [0, 0.0])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:40]
    per[name][0] += 1
    per[name][1] += (e - s) / 1e3
    ev.append((s, 1, r.get("Queue_Id", "")))
    ev.append((e, -1, r.get("Queue_Id", "")))
ev.sort()
depth = 0
last = ev[0][0]
busy = over = 0
for t, d, _ in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += d
    last = t
span = ev[-1][0] - ev[0][0]
print(f"kernels {len(rows)}  span {span/1e3:.0f} us  busy {busy/1e3:.0f} us ({100*busy/span:.0f}%)  >=2 in flight {over/1e3:.0f} us ({100*over/span:.0f}%)")
print("queues:", sorted({r.get('Queue_Id', '') for r in rows}))
for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:42s} n={n:5d} avg {us/n:8.2f} us  total {us:9.0f} us")
