#!/bin/bash
# GPU session r03-n: wide towers — the rank's ciphertexts bootstrapped in lockstep as one ciphertext of K-tower towers (fbb_bootstrap_wide)
# against the same ciphertexts over host threads: bytes, errors, bootstraps per second for groups of 4, 8, 16.
mkdir -p gpurun_out
G=$GRAFT_REPO_ROOT
B=$G/tests/hal/_build
export FHE_HIP_LIB=$G/openfhe-development_amd/csrc/libfhe_hip.so
cd /tmp && export TMPDIR=/tmp
FHE_HAL_REQUIRE_DEVICE=1 OMP_NUM_THREADS=8 timeout 1100 python3 - <<PY 2>&1 | grep -v "^InitPRNG" | tee $G/gpurun_out/wide_n.log
import sys, time
sys.path.insert(0, "$G")
from openfhe_amd import boot_batch as bb
K = 16
r = bb.run_rank(17, 65536, K, 8, 2, 0, "$B/libdetprng.so", warmup=1, key_threads=8, dump_path="/tmp/narrow.bin")
h = r.pop("handle")
print("over 8 host threads:", round(r["seconds_per_pass"], 4), "s per pass,", round(r["bootstraps_per_s"], 2), "bootstraps/s, max error", r["max_abs_error"])
for g in (8, 16, 4, 2, 1):
    t = h.bootstrap_wide(g, 2)
    print(f"wide, groups of {g}: {t:.4f} s per pass, {K / t:.2f} bootstraps/s, max error {max(h.check(i)[0] for i in range(K)):.3e}")
    if g == 8:
        h.dump("/tmp/wide.bin", 0, K)
        a, b = open("/tmp/narrow.bin", "rb").read(), open("/tmp/wide.bin", "rb").read()
        print("all", K, "bootstrapped ciphertexts:", len(a), "bytes,", "IDENTICAL to the narrow path's" if a == b else "DIFFERENT")
h.close()
PY
