#!/bin/bash
# round 5, call AP: how many workgroups a pair may grow to by adoption (ADOPT_GMAX 2 / 4 / 8) with the final kernel, and the iteration cutoff once more
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ap; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" -- "gmax4" "gmax8 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_gmax8.so" "gmax2 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_gmax2.so" "gmax4_kmax40 CVO_HIP_ADOPT_KMAX=40" "gmax8_kmax40 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_gmax8.so CVO_HIP_ADOPT_KMAX=40" | cut -c1-70
