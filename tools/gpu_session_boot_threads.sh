#!/bin/bash
# GPU session: config-4-size EvalBootstrap through the HIP backend of DCRTPoly for several OpenMP thread counts (pke's own
# parallel loops issue device operations from all of them).   usage: tools/gpu_session_boot_threads.sh [logN] [threads...]
LOGN=${1:-17}; shift
THREADS=${@:-"1 4 32"}
B=tests/hal/_build
mkdir -p gpurun_out
export FHE_HIP_LIB=$PWD/openfhe-development_amd/csrc/libfhe_hip.so FHE_HAL_REQUIRE_DEVICE=1
for t in $THREADS; do
  echo "== OMP_NUM_THREADS=$t"
  OMP_NUM_THREADS=$t timeout 900 $B/shim_ckks_hip /tmp/boot_hip_$t.bin $PWD/$B/libdetprng.so boottime $LOGN $((1 << (LOGN - 1))) 5 2>&1 | grep "rep \|bootstrap seconds\|per bootstrap\|keygen"
done 2>&1 | tee gpurun_out/boot_threads.log
