#!/bin/bash
# GPU session: BFV through the reference's CryptoContext on the HIP backend of DCRTPoly (tests + EvalMult timing at BASELINE
# configs[4]'s ring, N = 2^15, stock backend beside it).   usage: tools/gpu_session_bfv_shim.sh [depth] [reps]
DEPTH=${1:-10}; REPS=${2:-20}
B=tests/hal/_build
mkdir -p gpurun_out
export FHE_HIP_LIB=$PWD/openfhe-development_amd/csrc/libfhe_hip.so FHE_HAL_REQUIRE_DEVICE=1
{
echo "== gpu tests: shim (CKKS leveled, bootstrap, BFV x4 techniques), new C-ABI entries"
timeout 1500 python -m pytest tests/test_hal_shim.py tests/test_parity.py -q -m gpu -k "shim or inner_product or plus_minus" 2>&1 | tail -4
for t in BEHZ HPSPOVERQ; do
  echo "== BFV $t, N = 2^15, depth $DEPTH: stock backend (OMP_NUM_THREADS=32) then HIP backend"
  OMP_NUM_THREADS=32 timeout 900 $B/shim_ckks_stock /tmp/bfv_s.bin $PWD/$B/libdetprng.so bfv 15 $t $DEPTH 3 2>&1 | grep "^bfv"
  OMP_NUM_THREADS=32 timeout 900 $B/shim_ckks_hip /tmp/bfv_h.bin $PWD/$B/libdetprng.so bfv 15 $t $DEPTH $REPS 2>&1 | grep "^bfv\|^hal:"
  cmp /tmp/bfv_s.bin /tmp/bfv_h.bin && echo "BFV $t at 2^15: HIP backend == stock backend, bit for bit"
done
} 2>&1 | tee gpurun_out/session_bfv_shim.log
