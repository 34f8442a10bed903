#!/bin/bash
# round 4, session d: persistent row passes with the prefetch issued behind the tile's early vector-memory wait; the evidence tests
# after their fixes; per-member report of the reference's unit tests (CloneTowers on the device); bootstrap leg with the roofline counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NTT="--steps 10 --warmup 2 --no-cpu-baseline --no-evalmult --no-hadamard --no-bfv --no-lt --no-bootstrap --no-cc-evalmult"
for cfg in "pers8:" "pers16:FHE_NTT_PERS_STREAMS=16" "pers32:FHE_NTT_PERS_STREAMS=32" "plain:FHE_NTT_PERS_MIN_BATCH=0"; do
  n=${cfg%%:*}; e=${cfg#*:}
  env $e timeout 300 python bench.py $NTT > gpurun_out/r04_d_$n.json 2> gpurun_out/r04_d_$n.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_d_$n.json").read().strip().split("\n")[-1])
    print("$n", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("per_kernel_ms"), d.get("parity_at_full_size")[:40])
except Exception as e:
    print("$n failed", e, open("gpurun_out/r04_d_$n.err").read()[-800:])
PY
done
echo "== new tests"
timeout 1800 python -m pytest tests/test_multi_gpu_rccl_one_rank.py tests/test_multi_gpu_gloo.py tests/test_hal_shim.py tests/test_ref_unittests.py -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r04_d_newtests.txt
echo "== reference unit tests: per-member report"
(cd /tmp && FHE_HIP_LIB=$GRAFT_REPO_ROOT/openfhe-development_amd/csrc/libfhe_hip.so OMP_NUM_THREADS=8 timeout 600 $GRAFT_REPO_ROOT/tests/hal/_build/ut_hip "--gtest_filter=-*SERIALIZE*:UTBinInt.GetInternalRepresentation" 2>&1 | grep -E "tests ran|^hal|^halmember|^haldomain|^halcomposite|^haldecline" > $GRAFT_REPO_ROOT/gpurun_out/r04_d_ut_members.txt)
grep -E "tests ran|^hal:|haldomain|halcomposite|haldecline" gpurun_out/r04_d_ut_members.txt
echo "== bootstrap leg, 64 ciphertexts, lockstep groups of 32, roofline counters"
FHE_NTT_PERS_MIN_BATCH=0 timeout 900 python bench.py --batch 8 --steps 2 --warmup 1 --no-evalmult --no-hadamard --no-bfv --no-lt --no-cc-evalmult --no-cpu-baseline > gpurun_out/r04_d_boot64.json 2> gpurun_out/r04_d_boot64.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_d_boot64.json").read().strip().split("\n")[-1])
    print(json.dumps(d.get("evalbootstrap"), indent=1)[:3500])
except Exception as e:
    print("boot failed", e); print(open("gpurun_out/r04_d_boot64.err").read()[-1500:])
PY
