#!/bin/bash
# static-plan NTT kernels (pinned in-place butterflies): parity, bench, A/B against the run-time-plan kernel
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
echo "== bench static"; timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_r8.json; python -c "
import json;d=json.load(open('gpurun_out/bench_r8.json'));print(d['value'],d['ms_per_step'],d['roofline']);print(d['evalmult']);print(d.get('bfv_evalmult'))"
echo "== bench run-time plan (FHE_NTT_STATIC=0)"; FHE_NTT_STATIC=0 timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-evalmult 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print(d['value'],d['ms_per_step'],d['roofline'])"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r8 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-evalmult > $GRAFT_REPO_ROOT/gpurun_out/prof_r8.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_r8 -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-200
