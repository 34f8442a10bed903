#!/bin/bash
# runs tools/energybench with a rocm-smi sampler beside it; output: gpurun_out/r06_energybench.json + gpurun_out/r06_energybench_power.txt
mkdir -p gpurun_out
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" > gpurun_out/r06_energybench_idle.txt
( while true; do echo "t $(date +%s.%N)"; rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk"; sleep 0.15; done ) > gpurun_out/r06_energybench_power.txt &
SP=$!
timeout 300 tools/energybench > gpurun_out/r06_energybench.json
kill $SP
python3 - <<'PY'
import json, re
d = json.load(open("gpurun_out/r06_energybench.json"))
samples, t = [], None
for line in open("gpurun_out/r06_energybench_power.txt"):
    if line.startswith("t "):
        t = float(line.split()[1]); cur = {"t": t}; samples.append(cur)
    elif "Power" in line:
        cur["w"] = float(line.split(":")[-1])
    elif "sclk" in line:
        m = re.search(r"\((\d+)Mhz\)", line)
        if m: cur["mhz"] = int(m.group(1))
for r in d["results"] + d.get("memory", []):
    s = [x for x in samples if r["t_start"] + 0.6 <= x["t"] <= r["t_end"] and "w" in x]
    r["watts"] = round(sum(x["w"] for x in s) / len(s), 1) if s else None
    r["sclk_mhz"] = round(sum(x.get("mhz", 0) for x in s) / len(s)) if s else None
    print(r.get("instr") or r.get("what"), r.get("ginstr_per_s_per_simd_first"), r.get("ginstr_per_s_per_simd_settled"), r.get("GBps"), r["watts"], r["sclk_mhz"])
json.dump(d, open("gpurun_out/r06_energybench.json", "w"), indent=1)
PY
