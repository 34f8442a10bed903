#!/bin/bash
# fused (single-HBM-round-trip) NTT: parity under both hand-off paths, then A/B timing against the two-launch transform and
# the PMC traffic of the fused kernels.   usage: gpurun --timeout 1200 -- tools/gpu_fused_ab.sh
export FHE_BENCH_NO_TORCH=1
mkdir -p gpurun_out
NTT="--no-evalmult --no-bfv --no-hadamard --no-lt --no-cpu-baseline"
{
for m in 1 2 0; do
  echo "== parity FHE_NTT_FUSED=$m"
  FHE_NTT_FUSED=$m timeout 600 python -m pytest tests/test_parity.py tests/test_parity_full_shapes.py -q -m gpu -x -k "ntt or config1 or config2 or hybrid_keyswitch or config3 or rescale" 2>&1 | tail -2
done
for m in 0 1 1 0 2; do
  echo "== NTT leg FHE_NTT_FUSED=$m"
  FHE_NTT_FUSED=$m timeout 300 python bench.py $NTT 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   value', d['value'], 'ms/step', d['ms_per_step'], d['parity_at_full_size'][:60])"
done
echo "== EvalMult / BFV FHE_NTT_FUSED=0 vs 1"
for m in 0 1; do
  FHE_NTT_FUSED=$m timeout 400 python bench.py --batch 64 --steps 2 --warmup 1 --no-cpu-baseline --no-hadamard --no-lt 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   EvalMult', d['evalmult']['ops_per_s_per_gpu'], d['evalmult']['parity'][:20], ' BFV', d['bfv_evalmult']['ops_per_s_per_gpu'], d['bfv_evalmult']['with_relinearisation']['ops_per_s_per_gpu'])"
done
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/gpurun_out/pmc_fused_$c -- python $G/bench.py --steps 2 --warmup 1 --no-parity $NTT > $G/gpurun_out/pmc_fused_$c.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $G/gpurun_out/prof_fused_ntt -- python $G/bench.py --no-parity $NTT > $G/gpurun_out/prof_fused_ntt.log 2>&1
cd $G
python - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob(f"gpurun_out/pmc_fused_{c}/*/*counter_collection.csv")
    if not fs: continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"]==c: agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "ntt" in k: print(c,k,round(sum(v)/len(v)),"KiB  n=",len(v))
PY
f=$(ls -t gpurun_out/prof_fused_ntt/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-170
} 2>&1 | tee gpurun_out/fused_ab.log
