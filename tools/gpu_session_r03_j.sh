#!/bin/bash
# GPU session r03-j: the round's record (tools/gpu_record.sh r03: GPU tests, bench line, rocprofv3 kernel stats, PMC traffic and SQ
# passes) + the reference's unit tests with the trace + stability of the threaded cc->EvalMult figure + a kernel trace of the
# bootstrap batch over 8 streams (how much the streams overlap on the device).
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
bash tools/gpu_record.sh r03
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== reference unit tests with trace"
(time FHE_HAL_TRACE=1 OMP_NUM_THREADS=8 timeout 900 $B/ut_hip --gtest_filter="-*SERIALIZE*:UTBinInt.GetInternalRepresentation") > $G/gpurun_out/ut_trace_j.log 2>&1
grep "==========\|^hal\|^real\|FAILED" $G/gpurun_out/ut_trace_j.log | head
echo "== cc->EvalMult over 8 threads: 40 timed passes of 64 ciphertexts, three processes"
for i in 1 2 3; do
  OMP_NUM_THREADS=8 FHE_HAL_REQUIRE_DEVICE=1 timeout 300 $B/shim_ckks_hip /tmp/mb.bin $B/libdetprng.so multbatch 16 20 64 40 2>&1 | grep "multbatch seconds" | tee -a $G/gpurun_out/multbatch_j.log
done
echo "== bootstrap batch over 8 streams: kernel trace"
cat > /tmp/bb.py <<PY
import sys
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 8, 8, 1, 0, "$B/libdetprng.so", warmup=1, key_threads=8)
h = r.pop("handle")
print("8 threads", r["seconds_per_pass"], r["bootstraps_per_s"])
h.close()
PY
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r03j_bb -- python3 /tmp/bb.py > $G/gpurun_out/prof_r03j_bb.log 2>&1
grep "8 threads" $G/gpurun_out/prof_r03j_bb.log
python3 $G/tools/overlap_profile.py $(ls -t $G/gpurun_out/prof_r03j_bb/*/*kernel_trace.csv | head -1) 2>&1 | tee $G/gpurun_out/overlap_j.txt
rm -rf $G/gpurun_out/prof_r03j_bb
