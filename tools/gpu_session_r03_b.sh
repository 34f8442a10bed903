#!/bin/bash
# GPU session r03-b: the GPU test suite on the round-3 runtime (per-thread streams, member counters, chunked conversions), the
# reference's unit tests with FHE_HAL_TRACE, and the threaded cc->EvalMult batch (stream ordering under real asynchrony).
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== reference unit tests with trace"
(time FHE_HAL_TRACE=1 OMP_NUM_THREADS=8 timeout 1200 $G/tests/hal/_build/ut_hip --gtest_filter="-*SERIALIZE*:UTBinInt.GetInternalRepresentation") > $G/gpurun_out/ut_trace_b.log 2>&1
grep "==========\|^hal:\|^real\|FAILED" $G/gpurun_out/ut_trace_b.log | head
echo "== multbatch threads"
B=$G/tests/hal/_build
for T in 1 4 8 16; do
  OMP_NUM_THREADS=$T FHE_HAL_REQUIRE_DEVICE=1 timeout 600 $B/shim_ckks_hip /tmp/mb$T.bin $B/libdetprng.so multbatch 16 20 64 3 2>&1 | grep "multbatch seconds\|^hal:" | sed "s/^/T=$T /"
done
OMP_NUM_THREADS=8 timeout 900 $B/shim_ckks_stock /tmp/mbs.bin $B/libdetprng.so multbatch 16 20 64 1 2>&1 | grep "multbatch seconds"
cmp /tmp/mb8.bin /tmp/mbs.bin && echo "multbatch T=8 IDENTICAL to stock"; cmp /tmp/mb16.bin /tmp/mbs.bin && echo "multbatch T=16 IDENTICAL to stock"
echo "== bootstrap timing N=2^17"
OMP_NUM_THREADS=1 timeout 900 $B/shim_ckks_hip /tmp/bt.bin $B/libdetprng.so boottime 17 65536 3 2>&1 | grep "bootstrap seconds\|per bootstrap\|config4"
