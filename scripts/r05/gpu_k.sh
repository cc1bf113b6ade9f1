#!/bin/bash
# round 5, call K: config 5 (9.3 k points, three waves per SIMD, 37 blocks on 12 waves): the deal of the blocks with unequal shares for the early and late waves of a SIMD
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_k; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
L=$PWD/tmp_libs/libcvo_hip_skew.so
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "eth3d 16 4" -- "serpentine" "s0 CVO_HIP_LIB=$L CVO_HIP_WAVE_SKEW=0" "s10 CVO_HIP_LIB=$L CVO_HIP_WAVE_SKEW=0.10" "s20 CVO_HIP_LIB=$L CVO_HIP_WAVE_SKEW=0.20" "s30 CVO_HIP_LIB=$L CVO_HIP_WAVE_SKEW=0.30" | cut -c1-330
