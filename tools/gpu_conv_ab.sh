#!/bin/bash
# A/B of the conversion kernel's column-sum variants (FHE_CONV_SUM8) on the EvalMult / BFV legs
export FHE_BENCH_NO_TORCH=1
python -m pytest tests/test_parity.py tests/test_parity_bfv.py -q -m gpu -x -k "basis or switch or hybrid or behz or bfv or conv or expand" 2>&1 | tail -1
for v in 1 0 1 0; do
  echo "== FHE_CONV_SUM8=$v"
  FHE_CONV_SUM8=$v python bench.py --steps 2 --warmup 1 --batch 64 --no-cpu-baseline --no-hadamard --no-lt 2>&1 | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['evalmult']['ops_per_s_per_gpu'], d['bfv_evalmult']['ops_per_s_per_gpu'], d['bfv_evalmult']['with_relinearisation']['ops_per_s_per_gpu'])"
done
