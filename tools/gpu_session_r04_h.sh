#!/bin/bash
# round 4, session h: the BSGS baby steps in one inner-product launch (digits read once, c0*P in the store, the rotation as the
# plaintext-product kernel's gather) and the quotient-estimate canonicalisation of key-switch digits: parity on the GPU, then the lockstep
# bootstrap at 64 ciphertexts, its kernel census, and the EvalMult composite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== parity"
timeout 600 python -m pytest tests/test_parity_lt.py tests/test_parity_full_shapes.py tests/test_parity.py tests/test_multi_gpu_gloo.py -m gpu -q -x \
  -k "lt or bsgs or linear or keyswitch or key_switch or rotation or eval_mult or wide or hybrid" 2>&1 | tail -4 | tee gpurun_out/r04_h_tests.txt
timeout 400 python -m pytest tests/test_hal_shim.py -m gpu -q -x -k "bootstrap or leveled_ckks" 2>&1 | tail -4 | tee -a gpurun_out/r04_h_tests.txt
echo "== lockstep bootstrap, 64 ciphertexts"
timeout 300 python tools/boot_wide_profile.py sweep 64 16x2 32x1 8x4 2>&1 | grep -v "^Warning" | tee gpurun_out/r04_h_wide_sweep.txt
echo "== kernel census of the lockstep pass (one group of 32, one host thread)"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h -- python $GRAFT_REPO_ROOT/tools/boot_wide_profile.py run 32 32 2 > $GRAFT_REPO_ROOT/gpurun_out/r04_h_bootwide.log 2>&1 )
f=$(ls -t /tmp/prof_h/*/*kernel_trace.csv | head -1)
python tools/boot_wide_profile.py summarise $f 32 2 > gpurun_out/r04_h_bootstrap_wide_kernels.txt 2>&1
head -30 gpurun_out/r04_h_bootstrap_wide_kernels.txt
grep lockstep gpurun_out/r04_h_bootwide.log
echo "== EvalMult composite"
timeout 300 python bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-bfv --no-hadamard --no-lt 2>gpurun_out/r04_h_em.err | tail -1 > gpurun_out/r04_h_em.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_h_em.json").read())
print("evalmult", d["evalmult"]["ops_per_s_per_gpu"], d["evalmult"]["parity"][:60])
PY
