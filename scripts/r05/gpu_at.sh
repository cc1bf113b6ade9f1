#!/bin/bash
# round 5, call AT: the line search's table threshold (nonzeros per column above which the per-column table is made): 1 / 2 / 4 / 8
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_at; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" -- "factor4" "factor2 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_tab2.so" "factor8 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_tab8.so" "factor1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_tab1.so" | cut -c1-64
