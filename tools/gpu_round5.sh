#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_em -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --batch 64 --no-cpu-baseline --evalmult-batch 64 > $GRAFT_REPO_ROOT/gpurun_out/prof_em.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 gpurun_out/prof_em.log | cut -c1-1500
f=$(find gpurun_out/prof_em -name "*kernel_stats.csv" | head -1); cat $f | cut -c1-160
