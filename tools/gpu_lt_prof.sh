#!/bin/bash
# kernel-level profile of the BSGS linear-transform leg (batch 1 and 8)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export FHE_BENCH_NO_TORCH=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lt -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 32 --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard > $GRAFT_REPO_ROOT/gpurun_out/prof_lt.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_lt -name "*kernel_stats.csv" | head -1); head -30 $f | cut -c1-200
tail -1 gpurun_out/prof_lt.log | cut -c1-300
