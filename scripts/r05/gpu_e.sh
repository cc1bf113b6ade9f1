#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_e; mkdir -p $O
WAVES=1 CVO_HIP_WGS=1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_ktw.so timeout -k 10 200 python scripts/gpu_ktrace.py > $O/ktrace_waves.txt 2>&1; echo "rc=$?"
tail -50 $O/ktrace_waves.txt
