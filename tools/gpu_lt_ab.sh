#!/bin/bash
# A/B of the BSGS inner kernel variants (FHE_BSGS_CPL) on the linear-transform leg
export FHE_BENCH_NO_TORCH=1
python -m pytest tests/test_parity_lt.py -q -m gpu -x 2>&1 | tail -1
for cpl in 2 1; do
  echo "== FHE_BSGS_CPL=$cpl"
  FHE_BSGS_CPL=$cpl python bench.py --steps 1 --warmup 0 --batch 32 --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard 2>&1 | tail -1 | grep -o "\"per_batch\".*\"transforms_per_s_per_gpu\": [0-9.]*, \"cpu"
done
cd /tmp && export TMPDIR=/tmp
for cpl in 2 1; do
FHE_BSGS_CPL=$cpl timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lt_cpl$cpl -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 32 --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_lt_cpl$cpl -name "*kernel_stats.csv" | head -1); grep bsgs $f | cut -c1-160
done
