#!/bin/bash
# round 4, session b: what bounds the row passes now — ablations of the round-4 kernels (timing only, results wrong), occupancy
# sensitivity (LDS padding: 3 / 2 workgroups per CU), SQ counters of the headline leg; the whole GPU suite on the new kernels; the
# bootstrap leg at 64 ciphertexts per GPU with the live lockstep-vs-narrow comparison
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
NTT="--steps 5 --warmup 1 --no-cpu-baseline --no-evalmult --no-hadamard --no-bfv --no-lt --no-bootstrap --no-cc-evalmult --no-parity"
for n in hip notw nosync nolds nobfly occ3 occ2; do
  lib=tools/abl/libfhe_hip_$n.so; [ $n = hip ] && lib=openfhe-development_amd/csrc/libfhe_hip.so
  FHE_HIP_LIB=$PWD/$lib timeout 300 python bench.py $NTT > gpurun_out/r04_b_abl_$n.json 2> gpurun_out/r04_b_abl_$n.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_b_abl_$n.json").read().strip().split("\n")[-1])
    print("$n", d["ms_per_step"], (d.get("roofline") or {}).get("per_kernel_ms"))
except Exception as e:
    print("$n failed", e)
PY
done
echo "== GPU suite"; timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r04_b_gputests.txt
echo "== bootstrap leg, 64 ciphertexts"
timeout 900 python bench.py --batch 8 --steps 2 --warmup 1 --no-evalmult --no-hadamard --no-bfv --no-lt --no-cc-evalmult --cpu-seconds 2 > gpurun_out/r04_b_boot64.json 2> gpurun_out/r04_b_boot64.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_b_boot64.json").read().strip().split("\n")[-1])
    print(json.dumps(d.get("evalbootstrap"), indent=1)[:3000])
except Exception as e:
    print("boot failed", e); print(open("gpurun_out/r04_b_boot64.err").read()[-1500:])
PY
export FHE_BENCH_NO_TORCH=1
G=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $G/gpurun_out/r04_b_counters_avail.txt 2>&1
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
timeout 400 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $G/gpurun_out/pmc_r04b_sq_ntt -- python $G/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt --no-bootstrap --no-cc-evalmult > $G/gpurun_out/pmc_r04b_sq_ntt.log 2>&1
SQ2="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
timeout 400 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $G/gpurun_out/pmc_r04b_sq2_ntt -- python $G/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt --no-bootstrap --no-cc-evalmult > $G/gpurun_out/pmc_r04b_sq2_ntt.log 2>&1
cd $G; ls gpurun_out/pmc_r04b_sq_ntt/*/ | head
