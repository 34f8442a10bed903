#!/bin/bash
# round 4, session k = the round's record: the bench line exactly as the driver runs it, rocprofv3 kernel stats of the headline NTT leg and of
# the EvalMult leg, the PMC traffic passes of the headline leg (FETCH_SIZE, WRITE_SIZE: one pass each) and one pass of SQ counters over it;
# then `tools/collect_profiles.py r04` here.  (tools/gpu_record.sh is the same with the GPU suite and two more profiler passes in front.)
cd $GRAFT_REPO_ROOT
R=r04
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
NTT="--no-bootstrap --no-cc-evalmult --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt"
echo "== rocprof kernel stats, headline leg only"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $G/gpurun_out/prof_${R}_ntt -- python $G/bench.py $NTT > $G/gpurun_out/prof_${R}_ntt.log 2>&1
echo "== rocprof kernel stats, EvalMult leg at batch 256 (the NTT leg shrunk to 8 towers)"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $G/gpurun_out/prof_${R}_evalmult -- python $G/bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-bfv --no-hadamard --no-lt > $G/gpurun_out/prof_${R}_evalmult.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/gpurun_out/pmc_${R}_$c -- python $G/bench.py $NTT --steps 2 --warmup 1 > $G/gpurun_out/pmc_${R}_$c.log 2>&1
done
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
timeout 300 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $G/gpurun_out/pmc_${R}_sq_ntt -- python $G/bench.py $NTT --steps 2 --warmup 1 > $G/gpurun_out/pmc_${R}_sq_ntt.log 2>&1
cd $G
unset FHE_BENCH_NO_TORCH
python tools/collect_profiles.py $R --pmc-only   # (profiles/r04_pmc_*.json of THESE sources: the bench line below quotes them)
echo "== bench (default flags, as the driver runs it)"; ( time timeout 900 python bench.py 2>gpurun_out/bench_$R.err | tail -1 | tee gpurun_out/bench_$R.json | cut -c1-400 ) 2>&1 | tail -6
f=$(ls -t gpurun_out/prof_${R}_ntt/*/*kernel_stats.csv | head -1); head -6 $f | cut -c1-170
f=$(ls -t gpurun_out/prof_${R}_evalmult/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-170
# the traces themselves are large: keep the summaries only
find gpurun_out/prof_${R}_* gpurun_out/pmc_${R}_* -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
du -sh gpurun_out | tail -1
