#!/bin/bash
# A/B of kernel variants / ablation libraries on the lockstep bootstraps by KERNEL TIME (rocprofv3 --kernel-trace, one host thread: deterministic,
# no allocator noise).  Arguments: environment assignments per run, e.g. "FHE_KSM=0" "FHE_HIP_LIB=tools/abl6/libfhe_hip_x.so" (paths relative to the repo)
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1)); D=/tmp/ab_$i; rm -rf $D
  echo "== [$cfg]"
  cfg=${cfg//tools\/abl6/$G\/tools\/abl6}
  env $cfg timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python $G/tools/boot_wide_profile.py run 64 16 2 1 1 > $D.log 2>&1
  tail -1 $D.log | cut -c1-120
  f=$(ls -t $D/*/*kernel_trace.csv | head -1)
  python $G/tools/boot_wide_profile.py summarise $f 64 2 | grep -E "lockstep passes|ks_inner_multi|bsgs_inner" | cut -c1-150
done
