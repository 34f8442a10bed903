#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the kernels of the BSGS linear-transform leg
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lt_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --batch 32 --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard > $GRAFT_REPO_ROOT/gpurun_out/pmc_lt_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections,os
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob(f"gpurun_out/pmc_lt_{c}/*/*counter_collection.csv")
    f=max(fs,key=os.path.getmtime)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==c: agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        if "fhe::" in k: print(c,k,"n=",len(v),"max KiB=",round(max(v)),"min=",round(min(v)))
PY
