#!/bin/bash
# GPU session r03-h: host-side launch path.  One bootstrap is ~2000 launches of 10-50 us kernels and several host threads saturate near
# 55-70 k launches/s: HIP runtime knobs (kernel arguments in device memory, number of hardware queues) and the host-thread count of
# the two CryptoContext legs.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
L=$G/gpurun_out/launchpath_h.log
: > $L
run() {  # label, env...
  label=$1; shift
  echo "== $label" | tee -a $L
  env "$@" OMP_NUM_THREADS=1 timeout 600 $B/shim_ckks_hip /tmp/bt.bin $B/libdetprng.so boottime 17 65536 4 2>&1 | grep "bootstrap seconds\|per bootstrap: launches" | tee -a $L
  for T in 8 16; do
    env "$@" OMP_NUM_THREADS=$T FHE_HAL_REQUIRE_DEVICE=1 timeout 300 $B/shim_ckks_hip /tmp/mb.bin $B/libdetprng.so multbatch 16 20 64 5 2>&1 | grep "multbatch seconds" | sed "s/^/T=$T /" | tee -a $L
  done
}
run "default" FHE_DUMMY=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8
run "both" HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8
echo "== multbatch thread sweep (default environment)" | tee -a $L
for T in 4 12 24 32; do
  OMP_NUM_THREADS=$T FHE_HAL_REQUIRE_DEVICE=1 timeout 300 $B/shim_ckks_hip /tmp/mb.bin $B/libdetprng.so multbatch 16 20 64 5 2>&1 | grep "multbatch seconds" | sed "s/^/T=$T /" | tee -a $L
done
echo "== bootstrap batch: threads (fresh team each), default environment and both knobs" | tee -a $L
for E in "FHE_DUMMY=1" "HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=8"; do
env $E FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 900 python3 - <<PY 2>&1 | grep -v "^InitPRNG" | sed "s/^/[$E] /" | tee -a $L
import sys
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 8, 8, 2, 0, "$B/libdetprng.so", warmup=1, key_threads=8)
h = r.pop("handle")
print("8 threads", r["seconds_per_pass"], r["bootstraps_per_s"])
for T in (2, 3, 4, 6, 8):
    s = h.bootstrap_all(T, 2, 0)
    print(f"threads {T}: seconds per pass {s:.4f}  bootstraps/s {8 / s:.2f}")
h.close()
PY
done
