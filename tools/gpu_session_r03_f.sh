#!/bin/bash
# GPU session r03-f: after the hook-binding fix (KeySwitchCore + the bootstrap linear transforms really run as composites) and the fused
# rescale: the whole GPU test suite, the default bench line, bootstrap timing with the library's launch counters, a kernel trace of
# the bootstrap, a thread sweep over a batch of bootstraps.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
echo "== gpu tests"; (time timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -25) 2>&1
echo "== bench (default flags)"; (time timeout 1500 python bench.py 2>gpurun_out/bench_r03f.err | tail -1 > gpurun_out/bench_r03f.json) 2>&1 | grep real; cut -c1-1800 gpurun_out/bench_r03f.json; tail -5 gpurun_out/bench_r03f.err
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
echo "== bootstrap timing N=2^17 (one stream)"
OMP_NUM_THREADS=1 timeout 900 $B/shim_ckks_hip /tmp/bt.bin $B/libdetprng.so boottime 17 65536 5 2>&1 | grep "bootstrap seconds\|per bootstrap\|launches\|config4\|rep \|halcomposite\|keygen seconds\|differs" | tee $G/gpurun_out/boottime_f.log
echo "== the same with the unfused rescale"
FHE_RESCALE_UNFUSED=1 OMP_NUM_THREADS=1 timeout 900 $B/shim_ckks_hip /tmp/btu.bin $B/libdetprng.so boottime 17 65536 5 2>&1 | grep "bootstrap seconds\|per bootstrap: launches" | tee -a $G/gpurun_out/boottime_f.log
cmp /tmp/bt.bin /tmp/btu.bin && echo "fused and unfused rescale: bootstrapped ciphertext IDENTICAL" | tee -a $G/gpurun_out/boottime_f.log
echo "== bootstrap kernel trace"
OMP_NUM_THREADS=1 timeout 900 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r03f_boot -- $B/shim_ckks_hip /tmp/bt2.bin $B/libdetprng.so boottime 17 65536 3 > $G/gpurun_out/prof_r03f_boot.log 2>&1
python3 $G/tools/boot_profile.py $(ls -t $G/gpurun_out/prof_r03f_boot/*/*kernel_trace.csv | head -1) 3 2>&1 | head -40 | tee $G/gpurun_out/boot_profile_f.txt
rm -rf $G/gpurun_out/prof_r03f_boot
echo "== bootstrap batch N=2^17, 8 ciphertexts, threads sweep"
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1200 python3 - <<PY 2>&1 | grep -v "^InitPRNG" | tee $G/gpurun_out/bootbatch_f.log
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
