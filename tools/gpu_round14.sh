#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -2
echo "== bench"; timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_r14.json; python -c "
import json;d=json.load(open('gpurun_out/bench_r14.json'));print(d['value'],d['ms_per_step'],d['roofline']['per_kernel_ms']);print(d['evalmult']);print(d.get('bfv_evalmult'))"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r14 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --logn 16 --limbs 2 --batch 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_r14.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r14 -name "*kernel_stats.csv" | head -1); head -16 $f | cut -c1-180
