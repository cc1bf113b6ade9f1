#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_d; mkdir -p $O
for v in kt ktb; do CVO_HIP_WGS=1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$v.so timeout -k 10 200 python scripts/gpu_ktrace.py > $O/ktrace_$v.txt 2>&1; echo "$v rc=$?"; done
tail -3 $O/ktrace_kt.txt
