#!/bin/bash
# GPU session: BASELINE configs[2] (CKKS EvalMult + HYBRID key switch, N = 2^16, depth 20) through the reference's CryptoContext:
# a batch of ciphertexts spread over host threads, stock backend beside the HIP backend of DCRTPoly, products compared byte for byte.
#   usage: tools/gpu_session_multbatch.sh [B] [threads...]
BATCH=${1:-256}; shift
THREADS=${@:-"8 32"}
B=tests/hal/_build
mkdir -p gpurun_out
export FHE_HIP_LIB=$PWD/openfhe-development_amd/csrc/libfhe_hip.so FHE_HAL_REQUIRE_DEVICE=1
{
echo "== stock backend, 32 threads over the batch (64 ciphertexts)"
OMP_NUM_THREADS=32 timeout 900 $B/shim_ckks_stock /tmp/mb_s.bin $PWD/$B/libdetprng.so multbatch 16 20 64 1 2>&1 | grep "^multbatch"
for t in $THREADS; do
  echo "== HIP backend, OMP_NUM_THREADS=$t, $BATCH ciphertexts"
  OMP_NUM_THREADS=$t timeout 900 $B/shim_ckks_hip /tmp/mb_h_$t.bin $PWD/$B/libdetprng.so multbatch 16 20 $BATCH 3 2>&1 | grep "^multbatch\|^hal:"
done
echo "== byte comparison (64 ciphertexts, first and last product)"
OMP_NUM_THREADS=8 timeout 900 $B/shim_ckks_hip /tmp/mb_h.bin $PWD/$B/libdetprng.so multbatch 16 20 64 1 2>&1 | grep "^multbatch"
cmp /tmp/mb_s.bin /tmp/mb_h.bin && echo "EvalMult batch at 2^16 / depth 20: HIP backend == stock backend, bit for bit"
} 2>&1 | tee gpurun_out/session_multbatch.log
