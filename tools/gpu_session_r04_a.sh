#!/bin/bash
# round 4, session a: instruction-sequence costs (tools/seqbench.py), parity of the round-4 NTT kernels (truncated Shoup quotient,
# quotient-estimate reductions), A/B of the headline leg against the round-3 library (tools/abl/libfhe_hip_r3.so, built from 0848b24)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== seqbench"; timeout 300 python tools/seqbench.py run gpurun_out/r04_seqbench.json 2>&1 | tail -3
echo "== parity (NTT + composites through the C ABI)"
timeout 900 python -m pytest tests/test_parity.py tests/test_parity_full_shapes.py tests/test_parity_bfv.py tests/test_parity_lt.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r04_a_parity.txt
for lib in tools/abl/libfhe_hip_r3.so openfhe-development_amd/csrc/libfhe_hip.so; do
  n=$(basename $lib .so)
  echo "== headline leg with $n"
  FHE_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-evalmult --no-hadamard --no-bfv --no-lt --no-bootstrap --no-cc-evalmult > gpurun_out/r04_a_bench_$n.json 2> gpurun_out/r04_a_bench_$n.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_a_bench_$n.json").read().strip().split("\n")[-1])
print(d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("per_kernel_ms"), d.get("parity"))
PY
done
for lib in tools/abl/libfhe_hip_r3.so openfhe-development_amd/csrc/libfhe_hip.so; do
  n=$(basename $lib .so)
  echo "== EvalMult leg with $n"
  FHE_HIP_LIB=$PWD/$lib timeout 600 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-hadamard --no-bfv --no-lt --no-bootstrap --no-cc-evalmult > gpurun_out/r04_a_evalmult_$n.json 2> gpurun_out/r04_a_evalmult_$n.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_a_evalmult_$n.json").read().strip().split("\n")[-1])
print(d.get("evalmult"))
PY
done
