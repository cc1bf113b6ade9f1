#!/bin/bash
# round 5, call F: the deal of the row blocks with unequal shares for the early and the late waves of a SIMD (CVO_HIP_WAVE_SKEW)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_f; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
for sk in 0 0.11; do CVO_HIP_WAVE_SKEW=$sk WAVES=1 CVO_HIP_WGS=1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_ktw.so timeout -k 10 200 python scripts/gpu_ktrace.py > $O/ktrace_waves_$sk.txt 2>&1; echo "ktrace $sk rc=$? $(grep 'totals' $O/ktrace_waves_$sk.txt)"; done
CVO_HIP_WAVE_SKEW=0.1 timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py -x -q > $O/parity_skew.txt 2>&1; echo "parity skew rc=$? $(tail -1 $O/parity_skew.txt)"
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "tum 20 5" "tum 256 32" -- "s0 CVO_HIP_WAVE_SKEW=0" "s05 CVO_HIP_WAVE_SKEW=0.05" "s10 CVO_HIP_WAVE_SKEW=0.10" "s15 CVO_HIP_WAVE_SKEW=0.15" "s20 CVO_HIP_WAVE_SKEW=0.20" | cut -c1-260
