#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_tracker4; mkdir -p $O
for rep in 1 2; do for v in "CVO_HIP_FIRST_SCALE=1" "CVO_HIP_FIRST_SCALE=0.75" "CVO_HIP_FIRST_SCALE=0.5" "CVO_HIP_SKIN=0.03 CVO_HIP_SKIN_ALPHA=0.008" "CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.008" "CVO_HIP_PREDICT=1.0" "CVO_HIP_PREDICT=0"; do echo "== $v"; env $v timeout -k 10 200 python scripts/gpu_r4_tracker2.py 2>&1 | grep "queued score block on" | cut -c1-175; done; done | tee $O/narrow.txt
