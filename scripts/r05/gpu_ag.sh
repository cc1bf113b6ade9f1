#!/bin/bash
# round 5, call AG: the line search asks for its first records before its constants and table (their round trip under that work) against asking at the walk's start
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ag; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_config5.py tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" "eth3d 16 4" -- "early_ring" "prev CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-230
