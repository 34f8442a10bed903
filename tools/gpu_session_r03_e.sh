#!/bin/bash
# GPU session r03-e: the whole GPU test suite, the default bench line (as the driver runs it), the reference's unit tests with the
# per-kind host-mirror trace, a batch of bootstraps with a thread sweep.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
echo "== gpu tests"; timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -25
echo "== bench (default flags)"; (time timeout 1500 python bench.py 2>gpurun_out/bench_r03e.err | tail -1 > gpurun_out/bench_r03e.json) 2>&1 | grep real; cut -c1-1500 gpurun_out/bench_r03e.json; tail -5 gpurun_out/bench_r03e.err
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== reference unit tests with trace"
(time FHE_HAL_TRACE=1 OMP_NUM_THREADS=8 timeout 1200 $B/ut_hip --gtest_filter="-*SERIALIZE*:UTBinInt.GetInternalRepresentation") > $G/gpurun_out/ut_trace_e.log 2>&1
grep "==========\|^hal\|^real\|FAILED" $G/gpurun_out/ut_trace_e.log | head
echo "== bootstrap batch N=2^17, 8 ciphertexts, threads sweep"
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1500 python3 - <<PY 2>&1 | grep -v "^InitPRNG" | tee $G/gpurun_out/bootbatch_e.log
import sys, time
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 8, 4, 2, 0, "$B/libdetprng.so", warmup=1, key_threads=8)
h = r.pop("handle")
print("4 threads", {k: v for k, v in r.items()})
for T in (1, 1, 2, 3, 4, 6, 8):
    h.L.fbb_set_omp_threads(T)
    s = h.bootstrap_all(T, 2, 0)
    print(f"threads {T}: seconds per pass {s:.4f}  bootstraps/s {8 / s:.2f}")
print("max abs error", max(h.check(i)[0] for i in range(8)))
h.close()
PY
