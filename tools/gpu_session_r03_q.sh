#!/bin/bash
# GPU session r03-q: cc->EvalMult in lockstep (wide towers) at config 3's shape: GPU test against the stock backend, EvalMult per second.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
timeout 300 python -m pytest tests/test_hal_shim.py -m gpu -q -x -k "lockstep" 2>&1 | tail -3
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
for g in 64 128 32; do
  OMP_NUM_THREADS=1 FHE_HAL_REQUIRE_DEVICE=1 timeout 200 $B/shim_ckks_hip /tmp/mw$g.bin $B/libdetprng.so multbatch 16 20 256 10 $g 2>&1 | grep "multbatch seconds" | sed "s/^/lockstep groups of $g: /" | tee -a $G/gpurun_out/multwide_q.log
done
OMP_NUM_THREADS=8 FHE_HAL_REQUIRE_DEVICE=1 timeout 200 $B/shim_ckks_hip /tmp/mt.bin $B/libdetprng.so multbatch 16 20 256 10 2>&1 | grep "multbatch seconds" | sed "s/^/8 host threads: /" | tee -a $G/gpurun_out/multwide_q.log
cmp /tmp/mw64.bin /tmp/mt.bin && echo "lockstep and threaded products IDENTICAL" | tee -a $G/gpurun_out/multwide_q.log
