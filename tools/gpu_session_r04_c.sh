#!/bin/bash
# round 4, session c: persistent row passes with next-tile prefetch (3 workgroups per CU) against the plain kernels (same binary,
# FHE_NTT_PERS_MIN_BATCH=0 disables), streams-per-pair sweep; the new evidence tests (wide bootstrap on the GPU, RCCL in a world of one
# rank, set-up window of the shim tests, per-member report of the reference's unit tests); lockstep bootstrap groups of 32 / 16
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== persistent kernels: parity"; FHE_HIP_LIB=$PWD/openfhe-development_amd/csrc/libfhe_hip.so timeout 300 python tools/abl/pers_check.py
NTT="--steps 10 --warmup 2 --no-cpu-baseline --no-evalmult --no-hadamard --no-bfv --no-lt --no-bootstrap --no-cc-evalmult"
for cfg in "pers8:" "pers4:FHE_NTT_PERS_STREAMS=4" "pers16:FHE_NTT_PERS_STREAMS=16" "pers32:FHE_NTT_PERS_STREAMS=32" "plain:FHE_NTT_PERS_MIN_BATCH=0"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 300 python bench.py $NTT > gpurun_out/r04_c_$n.json 2> gpurun_out/r04_c_$n.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_c_$n.json").read().strip().split("\n")[-1])
    print("$n", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("per_kernel_ms"), d.get("parity_at_full_size"))
except Exception as e:
    print("$n failed", e, open("gpurun_out/r04_c_$n.err").read()[-800:])
PY
done
echo "== EvalMult leg, persistent / plain"
for cfg in "pers:" "plain:FHE_NTT_PERS_MIN_BATCH=0"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 400 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-hadamard --no-bfv --no-lt --no-bootstrap --no-cc-evalmult > gpurun_out/r04_c_em_$n.json 2> gpurun_out/r04_c_em_$n.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_c_em_$n.json").read().strip().split("\n")[-1])
    print("$n", d["evalmult"]["ops_per_s_per_gpu"], d["evalmult"]["parity"][:60])
except Exception as e:
    print("$n failed", e)
PY
done
echo "== new tests"
timeout 1500 python -m pytest tests/test_multi_gpu_rccl_one_rank.py tests/test_multi_gpu_gloo.py tests/test_hal_shim.py tests/test_parity_full_shapes.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r04_c_newtests.txt
echo "== reference unit tests: per-member report"
(cd /tmp && FHE_HIP_LIB=$GRAFT_REPO_ROOT/openfhe-development_amd/csrc/libfhe_hip.so OMP_NUM_THREADS=8 timeout 600 $GRAFT_REPO_ROOT/tests/hal/_build/ut_hip "--gtest_filter=-*SERIALIZE*:UTBinInt.GetInternalRepresentation" 2>&1 | grep -E "tests ran|^hal|^halmember|^haldomain|^halcomposite" > $GRAFT_REPO_ROOT/gpurun_out/r04_c_ut_members.txt)
grep -E "tests ran|^hal:|haldomain|halcomposite" gpurun_out/r04_c_ut_members.txt
for g in 32 16; do
  echo "== bootstrap leg, 64 ciphertexts, lockstep groups of $g"
  timeout 900 python bench.py --batch 8 --steps 2 --warmup 1 --no-evalmult --no-hadamard --no-bfv --no-lt --no-cc-evalmult --no-cpu-baseline --bootstrap-group $g > gpurun_out/r04_c_boot64_g$g.json 2> gpurun_out/r04_c_boot64_g$g.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_c_boot64_g$g.json").read().strip().split("\n")[-1])
    b = d.get("evalbootstrap"); print(b.get("bootstraps_per_s_per_gpu"), b.get("bootstraps_per_s_over_host_threads"), b.get("lockstep"))
except Exception as e:
    print("boot failed", e); print(open("gpurun_out/r04_c_boot64_g$g.err").read()[-1500:])
PY
done
