#!/bin/bash
# Round record on the GPU box: GPU tests, the bench line exactly as the driver runs it, rocprofv3 kernel stats of the same
# command (whole bench, headline NTT leg alone, EvalMult leg alone), the PMC traffic passes of the headline leg (FETCH_SIZE, WRITE_SIZE, one
# pass each) and one pass of SQ counters (VALU issue) over the headline and the EvalMult leg.
#   usage: gpurun --timeout 1500 -- tools/gpu_record.sh [tag] [notests]      (tag = r05 ...; then tools/collect_profiles.py tag)
# (round 4 ran the shorter tools/gpu_session_r04_k.sh: counter passes first, `collect_profiles.py tag --pmc-only` on the box, then the bench
# line — which so quotes the counters of the very sources it runs)
R=${1:-r05}
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  echo "== gpu tests"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
fi
echo "== bench (default flags, as the driver runs it)"; timeout 1200 python bench.py 2>gpurun_out/bench_$R.err | tail -1 | tee gpurun_out/bench_$R.json | cut -c1-600
export FHE_BENCH_NO_TORCH=1
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
echo "== rocprof kernel stats (same command as the bench line)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $G/gpurun_out/prof_$R -- python $G/bench.py --no-bootstrap --no-cc-evalmult --no-cpu-baseline --no-parity > $G/gpurun_out/prof_$R.log 2>&1
echo "== rocprof kernel stats, headline leg only (per-kernel averages = the B=1024 launches alone)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $G/gpurun_out/prof_${R}_ntt -- python $G/bench.py --no-bootstrap --no-cc-evalmult --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt > $G/gpurun_out/prof_${R}_ntt.log 2>&1
echo "== rocprof kernel stats, EvalMult leg at batch 256 (the NTT leg shrunk to 8 towers)"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $G/gpurun_out/prof_${R}_evalmult -- python $G/bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-bfv --no-hadamard --no-lt > $G/gpurun_out/prof_${R}_evalmult.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $G/gpurun_out/pmc_${R}_$c -- python $G/bench.py --no-bootstrap --no-cc-evalmult --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt > $G/gpurun_out/pmc_${R}_$c.log 2>&1
done
SQ="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $G/gpurun_out/pmc_${R}_sq_ntt -- python $G/bench.py --no-bootstrap --no-cc-evalmult --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-evalmult --no-bfv --no-hadamard --no-lt > $G/gpurun_out/pmc_${R}_sq_ntt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $G/gpurun_out/pmc_${R}_sq_evalmult -- python $G/bench.py --no-bootstrap --no-cc-evalmult --batch 8 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-bfv --no-hadamard --no-lt > $G/gpurun_out/pmc_${R}_sq_evalmult.log 2>&1
cd $G
f=$(ls -t gpurun_out/prof_${R}_ntt/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-170
f=$(ls -t gpurun_out/prof_${R}_evalmult/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-170
