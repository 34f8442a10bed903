#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for v in 0 1; do
echo "== bench FHE_NTT_LDS2=$v"; FHE_NTT_LDS2=$v timeout 900 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-evalmult 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print(d['value'],d['ms_per_step'],d['roofline']['per_kernel_ms'])"
done
