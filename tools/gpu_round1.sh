#!/bin/bash
# first GPU session: parity, ceilings, bench, profile.  Everything lands in gpurun_out/.
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== rocminfo"; rocminfo | grep -E "Marketing|Compute Unit|Max Clock" | head -6
echo "== nproc: $(nproc)"; lscpu | grep -E "Model name|^CPU\(s\)" 
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "== microbench"; timeout 300 ./tools/microbench 4 | tee gpurun_out/microbench.json
echo "== bench small"; timeout 600 python bench.py --steps 5 --warmup 2 --batch 128 --no-cpu-baseline --no-evalmult 2>&1 | tail -3 | tee gpurun_out/bench_b128.json
echo "== bench full"; timeout 900 python bench.py --steps 5 --warmup 2 --evalmult-batch 32 2>&1 | tail -3 | tee gpurun_out/bench_full.json
echo "== rocprof"; cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --batch 256 --no-cpu-baseline --no-evalmult > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); do head -12 $f; done
