#!/bin/bash
# ablation timing of the row pass: which resource bounds it?
export FHE_BENCH_NO_TORCH=1
for v in "" NOSYNC NOTWNOSYNC; do
  if [ -n "$v" ]; then export FHE_HIP_LIB=$PWD/tools/abl/libfhe_$v.so; fi
  echo "== variant '$v'"; timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-evalmult 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print(d['ms_per_step'],d['roofline']['per_kernel_ms'])"
done
