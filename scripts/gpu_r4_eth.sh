#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_eth; mkdir -p $O; hostname > $O/lease.txt
bash scripts/gpu_ab_env.sh $O/resort.txt 1 "eth3d 24 4" -- "head" "resort0 CVO_HIP_RESORT=0" "resort2 CVO_HIP_RESORT=2" "s12a01 CVO_HIP_SKIN=0.12 CVO_HIP_SKIN_ALPHA=0.01" "s12a01r0 CVO_HIP_SKIN=0.12 CVO_HIP_SKIN_ALPHA=0.01 CVO_HIP_RESORT=0" "head2"
