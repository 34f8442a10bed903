#!/bin/bash
# round 4, session g: lockstep groups spread over host threads (the GPU idles a fifth of a one-thread lockstep pass); an allocation that
# would eat into the reserve kept for kernel launches returns EVERY thread's cached buffers to the device first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 420 python tools/boot_wide_profile.py sweep 64 32x1 16x2 16x4 8x4 32x2 64x1 2>&1 | grep -v "^Warning" | tee gpurun_out/r04_g_wide_sweep.txt
