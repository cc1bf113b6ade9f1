#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_epi; mkdir -p $O
CVO_HIP_LIB=$GRAFT_REPO_ROOT/tmp_libs/libcvo_hip_${1:-epi}.so timeout -k 10 200 python scripts/gpu_r4_epi.py 2>&1 | grep -v amdgpu.ids | tee $O/epi_${1:-epi}.txt
