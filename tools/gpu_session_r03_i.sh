#!/bin/bash
# GPU session r03-i: buffers dropped by another thread go back to the stream that used them last (no cross-stream wait) against the
# former policy (FHE_HAL_FREE_TO_OWNER=0): repeated runs of the threaded cc->EvalMult leg (session h showed two modes, ~2.8 k and
# ~4.9 k EvalMult/s, for the same binary) and the bootstrap batch.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
L=$G/gpurun_out/free_to_owner_i.log
: > $L
for M in 1 0; do
  echo "== FHE_HAL_FREE_TO_OWNER=$M" | tee -a $L
  for T in 8 8 8 8 16 16 12; do
    FHE_HAL_FREE_TO_OWNER=$M OMP_NUM_THREADS=$T FHE_HAL_REQUIRE_DEVICE=1 timeout 300 $B/shim_ckks_hip /tmp/mb$M.bin $B/libdetprng.so multbatch 16 20 64 5 2>&1 | grep "multbatch seconds" | sed "s/^/T=$T /" | tee -a $L
  done
done
cmp /tmp/mb1.bin /tmp/mb0.bin && echo "products identical under both policies" | tee -a $L
for M in 1 0; do
FHE_HAL_FREE_TO_OWNER=$M FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 900 python3 - <<PY 2>&1 | grep -v "^InitPRNG" | sed "s/^/[to owner $M] /" | tee -a $L
import sys
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
r = bb.run_rank(17, 65536, 8, 8, 3, 0, "$B/libdetprng.so", warmup=1, key_threads=8)
h = r.pop("handle")
print("8 threads", r["seconds_per_pass"], r["bootstraps_per_s"])
for T in (4, 8, 8):
    s = h.bootstrap_all(T, 3, 0)
    print(f"threads {T}: seconds per pass {s:.4f}  bootstraps/s {8 / s:.2f}")
print("max abs error", max(h.check(i)[0] for i in range(8)))
h.close()
PY
done
