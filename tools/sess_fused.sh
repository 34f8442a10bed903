python -m pytest tests/test_parity.py tests/test_parity_full_shapes.py -m gpu -x -q -k "hybrid_keyswitch or mod_up or eval_mult or config3" 2>&1 | tail -4
EM="--no-bootstrap --no-cc-evalmult --no-bfv --no-hadamard --no-lt --no-cpu-baseline --steps 5 --warmup 2 --no-power"
for f in 1 0; do
  FHE_KS_FUSED_MODUP=$f FHE_BENCH_NO_TORCH=1 timeout 900 python bench.py $EM 2>gpurun_out/em_$f.err | tail -1 > gpurun_out/em_$f.json
  python3 -c "
import json;d=json.load(open('gpurun_out/em_$f.json'));e=d['evalmult'];print('fused=$f', e['ops_per_s_per_gpu'], e['ms_per_batch'], e['parity'][:60], e['roofline']['frac'], e['roofline']['moved_frac'])"
done
