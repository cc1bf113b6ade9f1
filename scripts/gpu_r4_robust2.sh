#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_robust; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_tail_scores.py -x -q > $O/pytest_head2.txt 2>&1; echo "HEAD rc=$? $(tail -1 $O/pytest_head2.txt)"
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_r03.so timeout -k 10 200 python -m pytest tests/test_gpu_tail_scores.py -q -k different_list_margins > $O/pytest_r03lib2.txt 2>&1; echo "round-3 library on the new tail test rc=$? (expected: failure) $(tail -3 $O/pytest_r03lib2.txt)"
