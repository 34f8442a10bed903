#!/bin/bash
# GPU session r03-m: kernel trace of the bootstrap batch over 8 streams (how much the streams overlap on the device).
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bb.py <<PY
import sys
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 8, 8, 1, 0, "$B/libdetprng.so", warmup=1, key_threads=8)
h = r.pop("handle")
print("8 threads", r["seconds_per_pass"], r["bootstraps_per_s"])
h.close()
PY
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r03m_bb -- python3 /tmp/bb.py > $G/gpurun_out/prof_r03m_bb.log 2>&1; echo "exit code $?"
grep -a "8 threads" $G/gpurun_out/prof_r03m_bb.log
python3 $G/tools/overlap_profile.py $(ls -t $G/gpurun_out/prof_r03m_bb/*/*kernel_trace.csv | head -1) 2>&1 | tee $G/gpurun_out/overlap_m.txt
gzip -c $(ls -t $G/gpurun_out/prof_r03m_bb/*/*kernel_trace.csv | head -1) > $G/gpurun_out/bb_kernel_trace_m.csv.gz; rm -rf $G/gpurun_out/prof_r03m_bb
