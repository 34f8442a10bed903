#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of a MULTI-stream run (a batch of bootstraps over host threads): for the last dense
window of the trace (everything after the last pause longer than `gap_ms`), the sum of kernel durations, the union of their
intervals (time during which at least one kernel ran), the span, and the distribution of concurrency (how much of the busy time had
1, 2, 3, ... kernels in flight), per queue launch counts.  usage: overlap_profile.py <kernel_trace.csv> [gap_ms=5]"""
import csv
import sys
from collections import Counter


def main():
    path = sys.argv[1]
    gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 5e6
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
    rows.sort()
    # windows separated by pauses longer than the gap; the last window that holds at least a quarter of the trace's kernel time (the timed
    # pass: what follows it is the decryption of the check)
    cuts, reach = [0], rows[0][1]
    for i in range(1, len(rows)):
        if rows[i][0] - reach > gap:
            cuts.append(i)
        reach = max(reach, rows[i][1])
    cuts.append(len(rows))
    wins = [rows[a:b] for a, b in zip(cuts, cuts[1:])]
    alltime = sum(e - s for s, e, _, _ in rows)
    big = [w for w in wins if sum(e - s for s, e, _, _ in w) >= alltime / 4] or [max(wins, key=len)]
    win = big[-1]
    print(f"{len(wins)} windows; kernel time per window (ms): {[round(sum(e - s for s, e, _, _ in w) / 1e6, 1) for w in wins]}")
    total = sum(e - s for s, e, _, _ in win)
    events = sorted([(s, 1) for s, _, _, _ in win] + [(e, -1) for _, e, _, _ in win])
    depth, last, hist = 0, events[0][0], Counter()
    for t, d in events:
        if depth > 0:
            hist[depth] += t - last
        depth += d
        last = t
    union = sum(hist.values())
    span = max(e for _, e, _, _ in win) - win[0][0]
    print(f"window: {len(win)} launches, span {span / 1e6:.2f} ms, sum of kernel durations {total / 1e6:.2f} ms, "
          f"at least one kernel running {union / 1e6:.2f} ms ({union / span:.3f} of the span), mean concurrency while busy {total / union:.2f}")
    for k in sorted(hist):
        print(f"  {k} kernel(s) in flight: {hist[k] / 1e6:8.2f} ms  ({hist[k] / union:.3f} of busy time)")
    q = Counter(r[2] for r in win)
    print("launches by queue:", dict(q))
    byk = Counter()
    for s, e, _, n in win:
        byk[n.split("(")[0].replace("void fhe::", "")[:60]] += e - s
    for n, t in byk.most_common(8):
        print(f"  {t / 1e6:8.2f} ms  {n}")


if __name__ == "__main__":
    main()
