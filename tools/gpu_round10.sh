#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
for t in 4 5 6 7 8; do
echo "== FHE_NTT_T1=$t"; FHE_NTT_T1=$t timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-evalmult 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.readline());print(d['value'],d['ms_per_step'],d['roofline']['per_kernel_ms'])"
done
echo "== parity at T1=6,8 (logN 16 only)"; for t in 6 8; do FHE_NTT_T1=$t timeout 600 python -m pytest tests/test_parity.py -m gpu -q -x -k "ntt_forward" 2>&1 | tail -1; done
