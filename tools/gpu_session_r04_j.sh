#!/bin/bash
# round 4, session j: the whole GPU suite on the round's final sources
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r04_j_gputests.txt
