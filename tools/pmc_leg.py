#!/usr/bin/env python3
"""Per-kernel HBM traffic and rate of one bench leg: two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE — separate runs, MI355X_MICROARCH.md:
FETCH_SIZE doubled on gfx950, units KiB) joined with the stand-alone kernel durations of the same runs (launches run one at a time under --pmc).
usage (GPU box): python tools/pmc_leg.py <tag> -- <bench.py flags>      output: gpurun_out/pmc_leg_<tag>.json + a table"""
import collections, csv, glob, json, os, subprocess, sys

tag = sys.argv[1]
flags = sys.argv[sys.argv.index("--") + 1:]
G = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
env = dict(os.environ, FHE_BENCH_NO_TORCH="1", TMPDIR="/tmp")
per = collections.defaultdict(lambda: collections.defaultdict(float))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    D = f"/tmp/pmcleg_{tag}_{c}"
    subprocess.run(["rm", "-rf", D])
    cmd = ["timeout", "600", "rocprofv3", "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", D, "--", sys.executable, os.path.join(G, "bench.py"), *flags]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    dur = {}
    for f in glob.glob(f"{D}/*/*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            dur[int(row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    for f in glob.glob(f"{D}/*/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            k = per[row["Kernel_Name"]]
            k[c] += float(row["Counter_Value"])
            k["calls_" + c] += 1
            k["ns_" + c] += dur.get(int(row["Dispatch_Id"]), 0)
table = []
for nm, k in per.items():
    fb, wb = k["FETCH_SIZE"] * 2048, k["WRITE_SIZE"] * 1024
    calls = max(k["calls_FETCH_SIZE"], k["calls_WRITE_SIZE"]) or 1
    ns = (k["ns_FETCH_SIZE"] + k["ns_WRITE_SIZE"]) / max(1, (k["calls_FETCH_SIZE"] > 0) + (k["calls_WRITE_SIZE"] > 0))
    table.append({"kernel": nm[:120], "calls": int(calls), "total_ms": round(ns / 1e6, 3), "us_per_call": round(ns / 1e3 / calls, 1),
                  "fetch_MB_per_call": round(fb / 1e6 / calls, 2), "write_MB_per_call": round(wb / 1e6 / calls, 2), "HBM_GBps": round((fb + wb) / ns, 1) if ns else None})
table.sort(key=lambda r: -r["total_ms"])
json.dump({"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py " + " ".join(flags), "by_kernel": table},
          open(os.path.join(G, "gpurun_out", f"pmc_leg_{tag}.json"), "w"), indent=1)
for r in table[:20]:
    print(f"{r['total_ms']:9.2f} ms {r['calls']:5d} x {r['us_per_call']:9.1f} us  fetch {r['fetch_MB_per_call']:9.1f} MB  write {r['write_MB_per_call']:9.1f} MB  {str(r['HBM_GBps']):>7} GB/s  {r['kernel'][:80]}")
