#!/bin/bash
# PMC evidence for the static NTT kernels: instruction counts, VALU activity, LDS conflicts (separate passes)
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"; do
  n=$(echo $set | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2_$n -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard > $GRAFT_REPO_ROOT/gpurun_out/pmc2_$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob,collections,os
res=collections.defaultdict(dict)
for d in glob.glob('gpurun_out/pmc2_*/'):
    fs=glob.glob(d+'*/*counter_collection.csv')
    if not fs: continue
    f=max(fs,key=os.path.getmtime)
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'ntt_static_kernel' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('<')[1].split('>')[0]
            agg[(k,r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k,c),v in agg.items(): res[k][c]=sum(v)/len(v)
for k,v in sorted(res.items()): print(k, {c:round(x) for c,x in sorted(v.items())})
PY
