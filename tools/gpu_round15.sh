#!/bin/bash
export FHE_BENCH_NO_TORCH=1
echo "== gpu ntt tests"; timeout 900 python -m pytest tests/test_parity.py -m gpu -q -x -k "ntt or config1" 2>&1 | tail -1
for i in 1 2; do
echo "== bench"; timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-evalmult 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print(d['value'],d['ms_per_step'],d['roofline']['per_kernel_ms'])"
done
