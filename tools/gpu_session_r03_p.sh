#!/bin/bash
# GPU session r03-p: the final state of the round — the whole GPU suite and the default bench line (with the lockstep bootstrap figure).
mkdir -p gpurun_out
echo "== gpu tests"; (time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) 2>&1
echo "== bench (default flags)"; (time timeout 900 python bench.py 2>gpurun_out/bench_r03p.err | tail -1 > gpurun_out/bench_r03p.json) 2>&1 | grep real
python3 - <<PY
import json
d = json.load(open("gpurun_out/bench_r03p.json"))
print("value", d["value"], "traffic", d["roofline"]["traffic"], "wasted", d["roofline"].get("wasted_traffic_ratio"), "binding", d["roofline"]["binding"].get("frac_of_issue_peak"))
print("evalbootstrap", json.dumps(d.get("evalbootstrap"))[:1500])
print("cc evalmult", (d.get("cryptocontext_evalmult") or {}).get("ops_per_s"), (d.get("cryptocontext_evalmult") or {}).get("parity"))
PY
tail -3 gpurun_out/bench_r03p.err
