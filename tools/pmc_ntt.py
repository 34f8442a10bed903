#!/usr/bin/env python3
"""rocprofv3 counter pass over the headline NTT leg (N=2^16, L=30, B=1024, 2 steps) -> per-kernel averages as JSON.
usage (GPU box): python tools/pmc_ntt.py <tag> <counter> [<counter> ...]   [env FHE_NTT_ROW8=... is inherited]
Output: gpurun_out/pmc_<tag>.json  (durations from the kernel trace of the same run; counter values are per-launch averages)"""
import collections, csv, glob, json, os, subprocess, sys

tag, counters = sys.argv[1], sys.argv[2:]
G = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
D = f"/tmp/pmc_{tag}"
env = dict(os.environ, FHE_BENCH_NO_TORCH="1", TMPDIR="/tmp")
leg = "--no-bootstrap --no-cc-evalmult --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt".split()
cmd = ["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", D, "--", sys.executable, os.path.join(G, "bench.py"), *leg]
r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
open(os.path.join(G, "gpurun_out", f"pmc_{tag}.log"), "w").write(r.stdout[-3000:] + "\n" + r.stderr[-3000:])
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{D}/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        per[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{D}/*/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
out = {"_how": " ".join(cmd[:-len(leg) - 2]) + " -- python bench.py " + " ".join(leg), "env_FHE_NTT_ROW8": os.environ.get("FHE_NTT_ROW8"), "kernels": {}}
for k, c in per.items():
    if "ntt_" not in k:
        continue
    name = k.replace("void fhe::", "").split("(fhe::")[0]
    d = {n: sum(v) / len(v) for n, v in c.items()}
    d["launches"] = max(len(v) for v in c.values())
    if dur.get(k):
        d["ms_under_counters"] = sum(dur[k]) / len(dur[k])
    out["kernels"][name] = d
json.dump(out, open(os.path.join(G, "gpurun_out", f"pmc_{tag}.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
