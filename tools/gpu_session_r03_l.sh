#!/bin/bash
# GPU session r03-l: one-stream bootstrap 48.2 ms in session k against 42.3 ms in sessions g / h: which of the follow-ups costs it?
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
L=$G/gpurun_out/boot_ab_l.log
: > $L
run() { label=$1; shift; echo "== $label" | tee -a $L
  env "$@" OMP_NUM_THREADS=1 timeout 600 $B/shim_ckks_hip /tmp/bt.bin $B/libdetprng.so boottime 17 65536 6 2>&1 | grep "bootstrap seconds\|rep \|per bootstrap: deviceOps" | tee -a $L; }
run "current" FHE_DUMMY=1
run "DCRTPoly = NativePoly on the host (FHE_HAL_ASSIGN_ON_HOST=1)" FHE_HAL_ASSIGN_ON_HOST=1
run "former free policy (FHE_HAL_FREE_TO_OWNER=0)" FHE_HAL_FREE_TO_OWNER=0
