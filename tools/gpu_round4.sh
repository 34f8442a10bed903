#!/bin/bash
mkdir -p gpurun_out
export FHE_BENCH_NO_TORCH=1
echo "== stagger sweep (T1=4 and 8)"
for st in 0 64 128 256; do FHE_NTT_STAGGER=$st timeout 300 python tools/ntt_sweep.py; done
for st in 0 128 256; do FHE_NTT_T1=8 FHE_NTT_STAGGER=$st timeout 300 python tools/ntt_sweep.py; done
