#!/bin/bash
# round 4: the depth-proportional list margin falling with ell (DevParams::alpha_gamma), one lease
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_gamma; mkdir -p $O; hostname > $O/lease.txt
CVO_HIP_ALPHA_GAMMA=1 timeout -k 10 300 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py -x -q 2>&1 | tail -1
bash scripts/gpu_ab_env.sh $O/sweep.txt 1 "tum 20 5" "tum 256 32" -- "g0" "g05 CVO_HIP_ALPHA_GAMMA=0.5" "g1 CVO_HIP_ALPHA_GAMMA=1" "g1a015 CVO_HIP_ALPHA_GAMMA=1 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.015" "g1a02 CVO_HIP_ALPHA_GAMMA=1 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.02" "g05a015 CVO_HIP_ALPHA_GAMMA=0.5 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.015" "g2a02 CVO_HIP_ALPHA_GAMMA=2 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.02" "g0b"
