#!/bin/bash
# GPU session r03-k: the backend after the record's follow-ups (thread exit without HIP calls, CloneTowers of host-only words on the
# host, DCRTPoly = NativePoly on the device): shim + unit-test GPU tests, the unit tests with the trace, bootstrap timing.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
echo "== shim + reference-unit-test gpu tests"; (time timeout 1500 python -m pytest tests/test_hal_shim.py tests/test_ref_unittests.py tests/test_multi_gpu_gloo.py -m gpu -q -x 2>&1 | tail -6) 2>&1
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== reference unit tests with trace"
(time FHE_HAL_TRACE=1 OMP_NUM_THREADS=8 timeout 900 $B/ut_hip --gtest_filter="-*SERIALIZE*:UTBinInt.GetInternalRepresentation") > $G/gpurun_out/ut_trace_k.log 2>&1
grep -a "==========\|^hal:\|^hal-other\|^halcomposite\|^real\|FAILED" $G/gpurun_out/ut_trace_k.log | head
echo "== bootstrap timing N=2^17 (one stream)"
OMP_NUM_THREADS=1 timeout 900 $B/shim_ckks_hip /tmp/bt.bin $B/libdetprng.so boottime 17 65536 5 2>&1 | grep "bootstrap seconds\|per bootstrap\|launches\|config4\|rep \|halcomposite\|halmemo\|keygen seconds\|differs" | tee $G/gpurun_out/boottime_k.log
echo "== the same under rocprofv3 --kernel-trace (thread exit inside the profiler)"
OMP_NUM_THREADS=4 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r03k -- $B/shim_ckks_hip /tmp/mb.bin $B/libdetprng.so multbatch 16 20 64 3 > $G/gpurun_out/prof_r03k.log 2>&1; echo "rocprofv3 multbatch exit code $?"
grep -a "multbatch seconds" $G/gpurun_out/prof_r03k.log
rm -rf $G/gpurun_out/prof_r03k
