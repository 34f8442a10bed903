#!/bin/bash
# round 4, session i: the multi-key inner product with every load of a coefficient issued up front and the next key's residues prefetched;
# single-key inner products (EvalMult's relinearisation, giant steps) through the same kernel (A/B: FHE_KS_INNER_VIA_MULTI=0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== parity"
timeout 400 python -m pytest tests/test_parity_lt.py tests/test_parity_full_shapes.py tests/test_parity.py -m gpu -q -x \
  -k "lt or bsgs or linear or keyswitch or key_switch or rotation or eval_mult or hybrid" 2>&1 | tail -3 | tee gpurun_out/r04_i_tests.txt
for v in 1 0; do
  echo "== EvalMult composite, FHE_KS_INNER_VIA_MULTI=$v"
  FHE_KS_INNER_VIA_MULTI=$v timeout 300 python bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-bfv --no-hadamard --no-lt 2>gpurun_out/r04_i_em$v.err | tail -1 > gpurun_out/r04_i_em$v.json
  python - <<PY
import json
d = json.loads(open("gpurun_out/r04_i_em$v.json").read())
print("evalmult", d["evalmult"]["ops_per_s_per_gpu"], d["evalmult"]["parity"][:60])
PY
done
echo "== lockstep bootstrap, 64 ciphertexts"
timeout 300 python tools/boot_wide_profile.py sweep 64 32x1 16x2 2>&1 | grep -v "^Warning" | tee gpurun_out/r04_i_wide_sweep.txt
echo "== the same with single-key inner products on the round-3 kernel"
FHE_KS_INNER_VIA_MULTI=0 timeout 300 python tools/boot_wide_profile.py sweep 64 32x1 2>&1 | grep -v "^Warning" | tee gpurun_out/r04_i_wide_sweep_single.txt
