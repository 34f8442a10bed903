#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV of `shim_ckks_hip ... boottime LOGN SLOTS REPS`: finds the kernel sequence of ONE
EvalBootstrap (the trace ends with REPS identical repetitions), and prints the GPU-busy time of that window next to its span
(first start .. last end), per-kernel totals and the launch count.  usage: boot_profile.py <kernel_trace.csv> [reps]"""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    import numpy as np

    ids = {}
    a = np.array([ids.setdefault(r[2], len(ids)) for r in rows], np.int32)
    n = len(a)
    st = np.array([r[0] for r in rows])
    en = np.array([r[1] for r in rows])
    # the trace ends with `reps` bootstraps followed by a short tail (file dump, decryption) that starts after a pause: cut it
    gaps = st[1:] - en[:-1]
    late = np.flatnonzero(gaps[-3000:] > 5_000_000)
    end = n - 3000 + int(late[-1]) + 1 if len(late) and n > 3000 else n
    # launches per bootstrap = the lag with the best match of the kernel-name sequence against itself (pke's OpenMP loops
    # issue operations from several threads, so consecutive bootstraps agree in most but not all positions)
    tail = a[max(0, end - 4 * (end // (reps + 2))):end]
    best, period = 0.0, None
    for k in range(500, len(tail) // 2):
        m = float((tail[k:] == tail[:-k]).mean())
        if m > best:
            best, period = m, k
    if period is None:
        print("no repeating block found; launches:", n)
        return
    print(f"period match {best:.3f}")
    rows = rows[:end]
    n = len(rows)
    win = rows[n - period:]
    busy = sum(e - s for s, e, _ in win)
    span = win[-1][1] - win[0][0]
    print(f"launches per bootstrap {period}   GPU busy {busy / 1e6:.2f} ms   span {span / 1e6:.2f} ms   busy/span {busy / span:.3f}")
    per = defaultdict(lambda: [0, 0])
    for s, e, nm in win:
        per[nm][0] += e - s
        per[nm][1] += 1
    print(f"{'ms':>9} {'calls':>6} {'avg us':>8}  kernel")
    for nm, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:25]:
        short = nm if len(nm) < 110 else nm[:107] + "..."
        print(f"{t / 1e6:9.3f} {c:6d} {t / c / 1e3:8.1f}  {short}")
    gaps = sorted((win[i + 1][0] - win[i][1]) for i in range(len(win) - 1))
    idle = sum(g for g in gaps if g > 0)
    print(f"idle between launches {idle / 1e6:.2f} ms; median gap {gaps[len(gaps) // 2] / 1e3:.1f} us; gaps > 100 us: {sum(1 for g in gaps if g > 100000)}")


if __name__ == "__main__":
    main()
