#!/bin/bash
# A/B of kernel variants on the lockstep bootstraps by KERNEL TIME (rocprofv3 --kernel-trace, one host thread: deterministic, no allocator noise)
cd /tmp && export TMPDIR=/tmp
G=$GRAFT_REPO_ROOT
i=0
for cfg in "$@"; do
  i=$((i+1)); D=/tmp/ab_$i; rm -rf $D
  echo "== [$cfg]"
  env $cfg timeout 600 rocprofv3 --kernel-trace --output-format csv -d $D -- python $G/tools/boot_wide_profile.py run 64 16 2 1 1 > $D.log 2>&1
  tail -1 $D.log | cut -c1-120
  f=$(ls -t $D/*/*kernel_trace.csv | head -1)
  python $G/tools/boot_wide_profile.py summarise $f 64 2 | grep -E "lockstep passes|ks_inner_multi|bsgs_inner|automorph" | cut -c1-150
done
