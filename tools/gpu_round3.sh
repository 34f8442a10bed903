#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== parity (ntt only)"; timeout 600 python -m pytest tests -m gpu -x -q -k "ntt or config1" 2>&1 | tail -3
echo "== fast kernel, T1 sweep, 4 WG/CU"
for t1 in 4 6 8; do FHE_NTT_T1=$t1 timeout 300 python tools/ntt_sweep.py; done | tee gpurun_out/sweep3_t1.jsonl
echo "== WG per CU sweep (T1=4)"
for w in 2 3 6 8; do FHE_NTT_WG_PER_CU=$w timeout 300 python tools/ntt_sweep.py; done | tee gpurun_out/sweep3_wg.jsonl
echo "== generic kernel for reference"
FHE_NTT_GENERIC=1 timeout 300 python tools/ntt_sweep.py
echo "== pmc"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc3.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/pmc3/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    if "ntt" in k: print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
