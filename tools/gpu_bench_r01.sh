#!/bin/bash
# round-1 record: bench line, rocprof kernel stats of the same command, PMC traffic passes
mkdir -p gpurun_out
echo "== gpu tests"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
echo "== bench (default flags, as the driver runs it)"; timeout 1200 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_r01.json | cut -c1-400
export FHE_BENCH_NO_TORCH=1
cd /tmp && export TMPDIR=/tmp
echo "== rocprof kernel stats (same command as the bench line)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01.log 2>&1
echo "== rocprof kernel stats, headline leg only (per-kernel averages = the B=1024 launches alone)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01_ntt -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard --no-lt > $GRAFT_REPO_ROOT/gpurun_out/prof_r01_ntt.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_r01_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard --no-lt > $GRAFT_REPO_ROOT/gpurun_out/pmc_r01_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/prof_r01_ntt/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-170
python - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"gpurun_out/pmc_r01_{c}/*/*counter_collection.csv")[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]==c: agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "ntt" in k: print(c,k,round(sum(v)/len(v)),"n=",len(v))
PY
