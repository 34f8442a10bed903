#!/bin/bash
# round 4, session f: lazily recorded weighted sums on the GPU (wide bootstrap tests, A/B of the bootstrap leg), the evidence tests with the
# allow-lists, kernel census of the lockstep bootstrap
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== tests"
timeout 1800 python -m pytest tests/test_multi_gpu_rccl_one_rank.py tests/test_multi_gpu_gloo.py tests/test_hal_shim.py tests/test_ref_unittests.py tests/test_parity.py -m gpu -q -k "not ntt" 2>&1 | tail -12 | tee gpurun_out/r04_f_tests.txt
for lz in 1 0; do
  echo "== bootstrap leg, 64 ciphertexts, groups of 32, FHE_HAL_LAZY_SUMS=$lz"
  FHE_HAL_LAZY_SUMS=$lz timeout 900 python bench.py --batch 8 --steps 2 --warmup 1 --no-evalmult --no-hadamard --no-bfv --no-lt --no-cc-evalmult --no-cpu-baseline > gpurun_out/r04_f_boot64_lazy$lz.json 2> gpurun_out/r04_f_boot64_lazy$lz.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r04_f_boot64_lazy$lz.json").read().strip().split("\n")[-1])
    b = d["evalbootstrap"]; print(b["bootstraps_per_s_per_gpu"], b["bootstraps_per_s_over_host_threads"], json.dumps(b["lockstep"])[:900])
except Exception as e:
    print("boot failed", e); print(open("gpurun_out/r04_f_boot64_lazy$lz.err").read()[-1500:])
PY
done
echo "== kernel census of the lockstep bootstrap"
G=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $G/gpurun_out/prof_r04f_bootwide -- python $G/tools/boot_wide_profile.py run 32 32 2 > $G/gpurun_out/r04_f_bootwide.log 2>&1
cd $G; tail -3 gpurun_out/r04_f_bootwide.log
f=$(ls -t gpurun_out/prof_r04f_bootwide/*/*kernel_trace.csv | head -1)
python tools/boot_wide_profile.py summarise $f 32 2 | tee gpurun_out/r04_bootstrap_wide_kernels.txt | head -40
rm -f $f  # (hundreds of MB: only the summary travels back)
