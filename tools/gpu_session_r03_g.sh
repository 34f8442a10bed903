#!/bin/bash
# GPU session r03-g: pair launches (both elements of a ciphertext per launch) and remembered rescales: shim + pair parity tests,
# bootstrap timing with the library's launch counters, a kernel trace of
# the bootstrap, a thread sweep over a batch of bootstraps.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
echo "== shim + pair gpu tests"; (time timeout 1500 python -m pytest tests/test_hal_shim.py tests/test_parity.py -m gpu -q -x -k "shim or pair or rescale" 2>&1 | tail -6) 2>&1
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== bootstrap timing N=2^17 (one stream)"
OMP_NUM_THREADS=1 timeout 900 $B/shim_ckks_hip /tmp/bt.bin $B/libdetprng.so boottime 17 65536 5 2>&1 | grep "bootstrap seconds\|per bootstrap\|launches\|config4\|rep \|halcomposite\|halmemo\|keygen seconds\|differs" | tee $G/gpurun_out/boottime_g.log
echo "== cc->EvalMult, 64 ciphertexts, N=2^16 depth 20, 8 threads"
OMP_NUM_THREADS=8 FHE_HAL_REQUIRE_DEVICE=1 timeout 600 $B/shim_ckks_hip /tmp/mb8.bin $B/libdetprng.so multbatch 16 20 64 5 2>&1 | grep "multbatch seconds\|halcomposite\|halmemo" | tee -a $G/gpurun_out/boottime_g.log
echo "== bootstrap kernel trace"
OMP_NUM_THREADS=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r03g_boot -- $B/shim_ckks_hip /tmp/bt2.bin $B/libdetprng.so boottime 17 65536 3 > $G/gpurun_out/prof_r03g_boot.log 2>&1
python3 $G/tools/boot_profile.py $(ls -t $G/gpurun_out/prof_r03g_boot/*/*kernel_trace.csv | head -1) 3 2>&1 | head -40 | tee $G/gpurun_out/boot_profile_g.txt
rm -rf $G/gpurun_out/prof_r03g_boot
echo "== bootstrap batch N=2^17, 8 ciphertexts, threads sweep"
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1200 python3 - <<PY 2>&1 | grep -v "^InitPRNG" | tee $G/gpurun_out/bootbatch_g.log
import sys, time
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 8, 4, 2, 0, "$B/libdetprng.so", warmup=1, key_threads=8)
h = r.pop("handle")
print("4 threads", {k: v for k, v in r.items()})
for T in (1, 2, 4, 8):
    h.L.fbb_set_omp_threads(T)
    s = h.bootstrap_all(T, 2, 0)
    print(f"threads {T}: seconds per pass {s:.4f}  bootstraps/s {8 / s:.2f}")
print("max abs error", max(h.check(i)[0] for i in range(8)))
h.close()
PY
