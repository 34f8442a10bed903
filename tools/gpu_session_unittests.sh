#!/bin/bash
# GPU session: the reference's own pke unit tests (1589 that need neither a serialisation library nor the reference tree at run
# time) on the HIP backend of DCRTPoly.   usage: tools/gpu_session_unittests.sh [gtest filter] [threads]
FILTER=${1:--*SERIALIZE*:UTBinInt.GetInternalRepresentation}
T=${2:-8}
mkdir -p gpurun_out
export FHE_HIP_LIB=$PWD/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp
(time OMP_NUM_THREADS=$T timeout 2400 $GRAFT_REPO_ROOT/tests/hal/_build/ut_hip --gtest_filter="$FILTER") > $GRAFT_REPO_ROOT/gpurun_out/ut_hip.log 2>&1
cd $GRAFT_REPO_ROOT
grep -c "^\[       OK \]" gpurun_out/ut_hip.log
grep "FAILED\|==========\|^hal:\|^real" gpurun_out/ut_hip.log | head -60
# the slowest tests
grep "^\[" gpurun_out/ut_hip.log | sed 's/.*(\([0-9]*\) ms)/\1 &/' | sort -rn | head -8 | cut -c1-160
