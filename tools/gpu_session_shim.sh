#!/bin/bash
# GPU session: new entry points + fused polynomial product + the HAL shim on the device (leveled CKKS, EvalBootstrap), then the
# config-4-size bootstrap (N = 2^17, 2^16 slots, {4,4}, SPARSE_TERNARY, FLEXIBLEAUTO) through the reference's own CryptoContext on
# the stock backend (host cores) and on the HIP backend.    usage: gpurun --timeout 1800 -- tools/gpu_session_shim.sh [boot_logn]
export FHE_BENCH_NO_TORCH=1
mkdir -p gpurun_out
LOGN=${1:-17}
B=tests/hal/_build
{
echo "== gpu tests (shim, poly_mul, new entry points)"
timeout 1200 python -m pytest tests/test_hal_shim.py tests/test_parity.py -m gpu -q -x -k "shim or poly_mul or mult_acc or switch_modulus_and or approx_mod_up or oversized or elementwise" 2>&1 | tail -3
if [ "$2" != "nobench" ]; then
echo "== bench: NTT leg + Hadamard / fused polynomial product"
timeout 600 python bench.py --no-evalmult --no-bfv --no-lt --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_polymul.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   value', d['value'], 'ms/step', d['ms_per_step']); print('  ', d['hadamard'])"
fi
echo "== config-4-size bootstrap, stock backend (OMP_NUM_THREADS=32)"
(time OMP_NUM_THREADS=32 timeout 900 $B/shim_ckks_stock gpurun_out/boot_stock.bin $PWD/$B/libdetprng.so boottime $LOGN) 2>&1 | grep -v "^dumped" | tail -9
echo "== config-4-size bootstrap, HIP backend of DCRTPoly (same program, same PRNG)"
(time OMP_NUM_THREADS=32 FHE_HAL_REQUIRE_DEVICE=1 timeout 900 $B/shim_ckks_hip gpurun_out/boot_hip.bin $PWD/$B/libdetprng.so boottime $LOGN $((1 << (LOGN - 1))) 3) 2>&1 | grep -v "^dumped" | cut -c1-220 | tail -22
cmp gpurun_out/boot_stock.bin gpurun_out/boot_hip.bin && echo "BOOTSTRAP at 2^$LOGN: HIP backend == stock backend, bit for bit"
rm -f gpurun_out/boot_stock.bin gpurun_out/boot_hip.bin
} 2>&1 | tee gpurun_out/session_shim.log
