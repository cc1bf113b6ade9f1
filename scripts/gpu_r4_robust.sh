#!/bin/bash
# round 4: the robustness changes -- tail agreement of members with different list margins (HEAD green, round-3 library must fail the same test),
# two-phase adoption accept with an injected helper loss, the gather that enters the collective with failure records
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_robust; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_tail_scores.py tests/test_gpu_adoption.py tests/test_gpu_multi.py tests/test_gpu_config3.py -x -q > $O/pytest_head.txt 2>&1; echo "HEAD rc=$? $(tail -1 $O/pytest_head.txt)"
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_r03.so timeout -k 10 120 python -m pytest tests/test_gpu_tail_scores.py -q -k different_list_margins > $O/pytest_r03lib.txt 2>&1; echo "round-3 library on the new tail test rc=$? (expected: failure) $(tail -1 $O/pytest_r03lib.txt)"
