#!/bin/bash
# GPU session r03-c: the composites behind pke (EvalMult = tensor + one composite key switch; bootstrapping's linear transforms = one
# call per level; HYBRID key generation on device towers) on the MI355X: shim GPU tests, threaded cc->EvalMult, bootstrap timing with
# a kernel trace, and the reference's unit tests with the host-mirror trace.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
echo "== shim gpu tests"; timeout 1200 python -m pytest tests/test_hal_shim.py -m gpu -q -x 2>&1 | tail -4
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== multbatch threads (N=2^16, depth 20, 64 ciphertexts)"
for T in 1 2 4 8 16; do
  OMP_NUM_THREADS=$T FHE_HAL_REQUIRE_DEVICE=1 timeout 600 $B/shim_ckks_hip /tmp/mb$T.bin $B/libdetprng.so multbatch 16 20 64 5 2>&1 | grep "multbatch seconds\|halcomposite" | sed "s/^/T=$T /"
done
OMP_NUM_THREADS=8 timeout 900 $B/shim_ckks_stock /tmp/mbs.bin $B/libdetprng.so multbatch 16 20 64 1 2>&1 | grep "multbatch seconds"
for T in 1 4 16; do cmp /tmp/mb$T.bin /tmp/mbs.bin && echo "multbatch T=$T IDENTICAL to stock"; done
echo "== bootstrap timing N=2^17"
OMP_NUM_THREADS=1 timeout 900 $B/shim_ckks_hip /tmp/bt.bin $B/libdetprng.so boottime 17 65536 5 2>&1 | grep "bootstrap seconds\|per bootstrap\|config4\|rep \|halcomposite\|keygen seconds" | tee $G/gpurun_out/boottime_c.log
echo "== bootstrap kernel trace"
OMP_NUM_THREADS=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r03_boot -- $B/shim_ckks_hip /tmp/bt2.bin $B/libdetprng.so boottime 17 65536 3 > $G/gpurun_out/prof_r03_boot.log 2>&1
python3 $G/tools/boot_profile.py $(ls -t $G/gpurun_out/prof_r03_boot/*/*kernel_trace.csv | head -1) 3 2>&1 | head -40 | tee $G/gpurun_out/boot_profile_c.txt
rm -rf $G/gpurun_out/prof_r03_boot
echo "== reference unit tests with trace"
(time FHE_HAL_TRACE=1 OMP_NUM_THREADS=8 timeout 1200 $B/ut_hip --gtest_filter="-*SERIALIZE*:UTBinInt.GetInternalRepresentation") > $G/gpurun_out/ut_trace_c.log 2>&1
grep "==========\|^hal:\|^real\|FAILED" $G/gpurun_out/ut_trace_c.log | head
