#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
export FHE_NTT_FULL=1
python tools/ntt_sweep.py
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc6a -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc6a.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc6b -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc6b.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc6c -- python $GRAFT_REPO_ROOT/tools/ntt_sweep.py > $GRAFT_REPO_ROOT/gpurun_out/pmc6c.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections
for d in ("pmc6a","pmc6b","pmc6c"):
    fs=glob.glob(f"gpurun_out/{d}/*/*counter_collection.csv")
    if not fs: print(d,"no output"); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:64]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if "ntt" in k and "false, false" in k: print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
