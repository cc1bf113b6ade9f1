#!/bin/bash
# round 5, call AF: wave issue priority by remaining work (s_setprio at block heads of the walks, by remaining records in the line search) against the hardware's oldest-first
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_af; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_waves.so timeout -k 10 300 python scripts/r05/probe_waves.py > $O/waves.txt 2>&1; grep -E "pair 0|^   1:|^   4:|^  12:|^  25:|^  40:" $O/waves.txt
timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_config5.py tests/test_gpu_adoption.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" "eth3d 16 4" -- "by_remaining" "oldest_first CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-330
