#!/bin/bash
# A/B of the two prepared (default-off) experiments on the GPU: parity subset under each knob, then EvalMult / BFV throughput.
#   FHE_CONV_SUM8=2     conversion / BEHZ column sums with 30-bit split factors
#   FHE_KS_FUSE_CONV=1  ModUp / ModDown conversions inside the forward NTT column pass
# usage: gpurun --timeout 900 -- tools/gpu_ab_experiments.sh
export FHE_BENCH_NO_TORCH=1
mkdir -p gpurun_out
run_bench() {
  python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline --no-hadamard --no-lt 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('   EvalMult', d['evalmult']['ops_per_s_per_gpu'], ' BFV', d['bfv_evalmult']['ops_per_s_per_gpu'], d['bfv_evalmult']['with_relinearisation']['ops_per_s_per_gpu'])"
}
{
echo "== parity under FHE_CONV_SUM8=2"
FHE_CONV_SUM8=2 timeout 600 python -m pytest tests/test_parity.py tests/test_parity_bfv.py tests/test_parity_lt.py -q -m gpu -x -k "basis or hybrid or behz or eval_mult or bsgs or linear" 2>&1 | tail -1
echo "== parity under FHE_KS_FUSE_CONV=1 (two-pass rings only take the fused path)"
FHE_KS_FUSE_CONV=1 timeout 600 python -m pytest tests/test_parity.py -q -m gpu -x -k "hybrid_keyswitch and (13-4 or 16-2 or 17-2)" 2>&1 | tail -1
FHE_KS_FUSE_CONV=1 PYTHONPATH=.:tests timeout 600 python tests/fused_conv_check.py openfhe-development_amd/csrc/libfhe_hip.so 2>&1 | tail -1
for cfg in "" "FHE_CONV_SUM8=2" "FHE_KS_FUSE_CONV=1" ""; do
  echo "== bench [$cfg]"
  env $cfg bash -c "$(declare -f run_bench); run_bench"
done
} 2>&1 | tee gpurun_out/ab_experiments.log
