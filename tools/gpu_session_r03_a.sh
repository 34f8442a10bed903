#!/bin/bash
# GPU session r03-a: (1) the reference's unit tests on the HIP backend with FHE_HAL_TRACE=1 (which members still run on the host
# mirror, by call site), (2) SQ counters (VALU issue) of the NTT leg and of the EvalMult leg on the current sources.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
(time FHE_HAL_TRACE=1 OMP_NUM_THREADS=8 timeout 1200 $G/tests/hal/_build/ut_hip --gtest_filter="-*SERIALIZE*:UTBinInt.GetInternalRepresentation") > $G/gpurun_out/ut_trace.log 2>&1
grep "==========\|^hal:\|^real" $G/gpurun_out/ut_trace.log | head
export FHE_BENCH_NO_TORCH=1
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $G/gpurun_out/pmc_r03_sq_ntt -- python $G/bench.py --no-bootstrap --no-cc-evalmult --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt > $G/gpurun_out/pmc_r03_sq_ntt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $G/gpurun_out/pmc_r03_sq_evalmult -- python $G/bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-bfv --no-hadamard --no-lt > $G/gpurun_out/pmc_r03_sq_evalmult.log 2>&1
cd $G
ls gpurun_out/pmc_r03_sq_ntt/*/ gpurun_out/pmc_r03_sq_evalmult/*/ 2>&1 | head
tail -3 gpurun_out/pmc_r03_sq_ntt.log
