#!/bin/bash
# line-search record ring depth (CVO_LS_RD: steps of nonzero records in flight per lane): 4 (default) against 2 / 6 / 8 (experiment builds)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_lsrd; mkdir -p $O; hostname > $O/lease.txt
L=$GRAFT_REPO_ROOT/tmp_libs
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "tum 64 8" -- "rd4" "rd2 CVO_HIP_LIB=$L/libcvo_hip_ls2.so" "rd6 CVO_HIP_LIB=$L/libcvo_hip_ls6.so" "rd8 CVO_HIP_LIB=$L/libcvo_hip_ls8.so"
