#!/bin/bash
# round 5, call AS: the cull with single 64-row units ALWAYS (tighter boxes, half the arithmetic per unit, twice the units) against block pairs where they are plentiful
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_as; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" "eth3d 16 4" -- "pairs_where_plentiful" "always_single CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_split2.so" | cut -c1-200
