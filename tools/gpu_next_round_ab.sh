#!/bin/bash
# First GPU session of the next round: parity + A/B of the experiments prepared (emulator-verified only) at the end of round 1.
#   FHE_CONV_SUM8=2     conversion / BEHZ column sums with 30-bit split factors (83 vs 103 VALU per output)
#   FHE_KS_FUSE_CONV=1  ModUp / ModDown conversions inside the forward NTT column pass
# usage: gpurun --timeout 900 -- tools/gpu_next_round_ab.sh
export FHE_BENCH_NO_TORCH=1
run_bench() {
  python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline --no-hadamard 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   EvalMult', d['evalmult']['ops_per_s_per_gpu'], ' BFV', d['bfv_evalmult']['ops_per_s_per_gpu'], d['bfv_evalmult']['with_relinearisation']['ops_per_s_per_gpu'], ' LT ms', d['linear_transform']['per_batch']['1']['ms_per_batch'], d['linear_transform']['per_batch']['8']['ms_per_batch'])"
}
echo "== parity under FHE_CONV_SUM8=2"
FHE_CONV_SUM8=2 python -m pytest tests/test_parity.py tests/test_parity_bfv.py tests/test_parity_lt.py -q -m gpu -x -k "basis or hybrid or behz or eval_mult or bsgs or linear" 2>&1 | tail -1
echo "== parity under FHE_KS_FUSE_CONV=1 (two-pass rings only take the fused path)"
FHE_KS_FUSE_CONV=1 python -m pytest tests/test_parity.py -q -m gpu -x -k "hybrid_keyswitch and (13-4 or 16-2 or 17-2)" 2>&1 | tail -1
FHE_KS_FUSE_CONV=1 PYTHONPATH=.:tests python tests/fused_conv_check.py openfhe-development_amd/csrc/libfhe_hip.so 2>&1 | tail -1
for cfg in "" "FHE_CONV_SUM8=2" "FHE_KS_FUSE_CONV=1" "FHE_CONV_SUM8=2 FHE_KS_FUSE_CONV=1" ""; do
  echo "== bench [$cfg]"
  env $cfg bash -c "$(declare -f run_bench); run_bench"
done
