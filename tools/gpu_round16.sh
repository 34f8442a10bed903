#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== gpu tests (new)"; timeout 900 python -m pytest tests -m gpu -q -x -k "rotations or mod_reduce or expand or keyswitch" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r16 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --logn 16 --limbs 2 --batch 8 > $GRAFT_REPO_ROOT/gpurun_out/prof_r16.log 2>&1
cd $GRAFT_REPO_ROOT
grep -a evalmult gpurun_out/prof_r16.log | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print(d['evalmult'])"
python - <<'PY'
import csv,glob,os
f=max(glob.glob('gpurun_out/prof_r16/*/*kernel_trace.csv'),key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last EvalMult iteration: find last tensor kernel, print until end
idx=[i for i,r in enumerate(rows) if 'tensor_kernel' in r['Kernel_Name']]
i0=idx[-1]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:]:
    print(r['Kernel_Name'][:58].ljust(58),(int(r['Start_Timestamp'])-t0)//1000,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))//1000)
PY
