#!/bin/bash
# samples power / clocks while the headline NTT leg runs (measurement tool): is the leg power-capped?
# usage (GPU box): tools/power_probe.sh [env assignments...]   output: gpurun_out/power_probe.txt
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out
OUT=gpurun_out/power_probe.txt
{ rocm-smi --showmaxpower --showperflevel 2>&1 | grep -v "^=" | head -20; } > $OUT
NTT_ONLY="--no-bootstrap --no-cc-evalmult --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard --no-lt --no-parity"
FHE_BENCH_NO_TORCH=1 python bench.py $NTT_ONLY --steps 900 --warmup 3 > gpurun_out/power_probe_bench.json 2>gpurun_out/power_probe_bench.err &
BP=$!
sleep 4
for i in $(seq 1 40); do
  echo "--- sample $i $(date +%s.%N)" >> $OUT
  rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|memory)" >> $OUT
  sleep 0.7
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -1 gpurun_out/power_probe_bench.json | cut -c1-300 >> $OUT
cat $OUT
