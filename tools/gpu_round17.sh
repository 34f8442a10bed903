#!/bin/bash
export FHE_BENCH_NO_TORCH=1
echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x -k "rotations or keyswitch or ntt" 2>&1 | tail -2
for i in 1 2; do
timeout 600 python bench.py --steps 9 --warmup 2 --no-cpu-baseline --logn 16 --limbs 2 --batch 8 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print(d['evalmult'])"
done
