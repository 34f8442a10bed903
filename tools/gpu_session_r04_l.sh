#!/bin/bash
# round 4, sessions l and o (o = the same after the best-fit buffer caches, the stream-sync drains and the resident cc->EvalMult figure): the bench line again as the driver runs it, after the bootstrap leg learnt to hand its buffer caches, keys and contexts
# back to the device when it closes (session k: the cc->EvalMult leg, a child process, found the device full: 353 op/s, lockstep out of memory)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python bench.py 2>gpurun_out/bench_r04.err | tail -1 | tee gpurun_out/bench_r04.json | cut -c1-300 ) 2>&1 | tail -6
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r04.json").read())
print("value", d["value"], d["ms_per_step"], "traffic", d["roofline"]["traffic"])
print("evalmult", d["evalmult"]["ops_per_s_per_gpu"])
print("boot", d["evalbootstrap"]["bootstraps_per_s_per_gpu"], d["evalbootstrap"]["lockstep"].get("parity", "")[:40])
print("ccm", json.dumps(d["cryptocontext_evalmult"])[:900])
PY
