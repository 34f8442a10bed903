#!/bin/bash
# GPU session r03-d: (1) the shim GPU tests with full failure output, (2) a batch of bootstraps at N = 2^17 over host threads (one
# stream per thread), (3) threaded cc->EvalMult with KeySwitchCore as a composite, (4) the reference's unit tests with the per-kind trace.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
echo "== shim gpu tests"; timeout 1500 python -m pytest tests/test_hal_shim.py -m gpu -q -x 2>&1 | tail -30
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== bootstrap batch N=2^17, 8 ciphertexts, threads sweep"
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1500 python3 - <<PY 2>&1 | grep -v "^InitPRNG" | tee $G/gpurun_out/bootbatch_d.log
import sys, time
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 8, 1, 1, 0, "$B/libdetprng.so", warmup=1)
h = r.pop("handle")
print("single thread", {k: v for k, v in r.items()})
for T in (2, 4, 8):
    s = h.bootstrap_all(T, 2, 0)
    print(f"threads {T}: seconds per pass {s:.4f}  bootstraps/s {8 / s:.2f}")
print("max abs error", max(h.check(i)[0] for i in range(8)))
PY
echo "== multbatch threads (N=2^16, depth 20, 64 ciphertexts)"
for T in 1 4 8 16; do
  OMP_NUM_THREADS=$T FHE_HAL_REQUIRE_DEVICE=1 timeout 600 $B/shim_ckks_hip /tmp/mb$T.bin $B/libdetprng.so multbatch 16 20 64 5 2>&1 | grep "multbatch seconds\|halcomposite" | sed "s/^/T=$T /"
done
echo "== reference unit tests with trace"
(time FHE_HAL_TRACE=1 OMP_NUM_THREADS=8 timeout 1200 $B/ut_hip --gtest_filter="-*SERIALIZE*:UTBinInt.GetInternalRepresentation") > $G/gpurun_out/ut_trace_d.log 2>&1
grep "==========\|^hal:\|^real\|FAILED" $G/gpurun_out/ut_trace_d.log | head
