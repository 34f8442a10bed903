#!/usr/bin/env python3
"""Compares two phases of one process in a rocprofv3 kernel trace (CSV): launches / time per kernel in either phase.  The phases are
delimited by the launches of a marker kernel: phase A = from marker launch number `skip` + 1 up to (not including) launch `skip` + `count`
+ 1, phase B = the `count` marker launches after that (measurement aid for tests/hal/shim_ckks.cpp multbatch: marker tensor_kernel,
skip 1 (the narrow warm-up), count = passes x groups: the packed lockstep passes against the resident ones).
   usage: trace_halves.py <kernel_trace.csv> <marker substring> <skip> <count>"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marker, skip, count = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
marks = [i for i, r in enumerate(rows) if marker in r[2]]
a0, b0 = marks[skip], marks[skip + count]
b1 = marks[skip + 2 * count] if len(marks) > skip + 2 * count else len(rows)
for name, part in (("phase A", rows[a0:b0]), ("phase B", rows[b0:b1])):
    per = defaultdict(lambda: [0, 0])
    for s, e, k in part:
        per[k][0] += e - s
        per[k][1] += 1
    busy = sum(v[0] for v in per.values())
    print(f"== {name}: {len(part)} launches, busy {busy / 1e6:.1f} ms, span {(part[-1][1] - part[0][0]) / 1e6:.1f} ms")
    for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {t / 1e6:9.2f} ms {c:6d} calls {t / c / 1e3:9.1f} us  {k[:100]}")
    # where the device idles: pauses longer than 20 us by (kernel before, kernel after)
    idle = defaultdict(lambda: [0, 0])
    for (s0, e0, k0), (s1, e1, k1) in zip(part, part[1:]):
        if s1 - e0 > 20000:
            key = (k0.split("(")[0][-60:], k1.split("(")[0][-60:])
            idle[key][0] += s1 - e0
            idle[key][1] += 1
    for (k0, k1), (t, c) in sorted(idle.items(), key=lambda kv: -kv[1][0])[:8]:
        print(f"  idle {t / 1e6:8.2f} ms in {c:4d} pauses between {k0}  ->  {k1}")
if len(sys.argv) > 5:  # detail: every launch between marker launch n and n + 5, with the pause before it (us)
    n = int(sys.argv[5])
    seg = rows[marks[n] - 3:marks[n + 5]]
    for (s0, e0, k0), (s1, e1, k1) in zip(seg, seg[1:]):
        print(f"  +{(s1 - e0) / 1e3:9.1f} us idle, then {(e1 - s1) / 1e3:8.1f} us  {k1.split('(')[0][-70:]}")
