NTT_ONLY="--no-bootstrap --no-cc-evalmult --no-cpu-baseline --no-evalmult --no-bfv --no-hadamard --no-lt --no-parity"
for ln in 13 14 15 17; do
  b=$((1024 * 65536 / (1 << ln)))
  for m in 0 1; do
    FHE_NTT_ROW8=$m FHE_BENCH_NO_TORCH=1 timeout 600 python bench.py $NTT_ONLY --logn $ln --batch $b --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/ring_${ln}_$m.json
    python3 -c "
import json;d=json.load(open('gpurun_out/ring_${ln}_$m.json'));print($ln,$m,d['ms_per_step'],d['roofline']['per_kernel_ms'])"
  done
done
