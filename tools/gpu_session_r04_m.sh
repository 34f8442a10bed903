#!/bin/bash
# round 4, session m: released buffers serve requests of smaller size classes too (best fit up to 4x): does the working set of more
# ciphertexts in flight fit, and does the first lockstep setting after the threaded pass still pay for the reclamation?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 240 python tools/boot_wide_profile.py sweep 64 16x2 32x1 16x4 32x2 2>&1 | grep -v "^Warning" | tee gpurun_out/r04_m_wide_sweep.txt
