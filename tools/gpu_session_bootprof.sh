#!/bin/bash
# GPU session: rocprofv3 kernel trace of the config-4-size EvalBootstrap through the HIP backend of DCRTPoly, reduced to the
# kernels of one bootstrap (tools/boot_profile.py).   usage: tools/gpu_session_bootprof.sh [logN] [threads]
LOGN=${1:-17}
B=tests/hal/_build
mkdir -p gpurun_out
export TMPDIR=/tmp FHE_HIP_LIB=$PWD/openfhe-development_amd/csrc/libfhe_hip.so
OUT=$PWD/gpurun_out/prof_boot
rm -rf $OUT
(cd /tmp && OMP_NUM_THREADS=32 FHE_HAL_REQUIRE_DEVICE=1 timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o boot -- \
   $OLDPWD/$B/shim_ckks_hip /tmp/boot_hip.bin $OLDPWD/$B/libdetprng.so boottime $LOGN $((1 << (LOGN - 1))) 3) 2>&1 | grep -v "^dumped" | grep -v "simple_timer" | cut -c1-200 | tail -12
CSV=$(find $OUT -name "*kernel_trace.csv" | head -1)
python tools/boot_profile.py $CSV 3 | tee gpurun_out/boot_profile.txt
gzip -c $CSV > gpurun_out/boot_kernel_trace.csv.gz; find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/boot_kernel_stats.csv
