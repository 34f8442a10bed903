#!/bin/bash
# round 4, session n: the cc->EvalMult leg of bench.py with the lockstep multiplications also timed on operands that stay wide
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 python bench.py --no-bootstrap --no-evalmult --no-bfv --no-hadamard --no-lt --no-cpu-baseline --no-parity --batch 8 --steps 1 --warmup 0 2>gpurun_out/r04_n.err | tail -1 > gpurun_out/r04_n.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_n.json").read())
print(json.dumps(d["cryptocontext_evalmult"])[:1500])
PY
tail -3 gpurun_out/r04_n.err
