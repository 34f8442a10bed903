#!/bin/bash
# BFV EvalMult composite + conversion-kernel scalar-table change: parity, EvalMult legs, kernel stats
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3
echo "== bench evalmult + bfv"; timeout 900 python bench.py --steps 6 --warmup 1 2>&1 | tail -1 > gpurun_out/bench_r7.json; python -c "
import json;d=json.load(open('gpurun_out/bench_r7.json'));print(d['value'],d['evalmult']);print(d.get('bfv_evalmult'))"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r7 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r7.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r7 -name "*kernel_stats.csv" | head -1); head -24 $f | cut -c1-200
