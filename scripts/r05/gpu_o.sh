#!/bin/bash
# round 5, call O: correctly rounded logarithm in the gates (device, host, oracle): the pair-value and float-routine KATs, the parity suites
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_o; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_pair_values.py tests/test_gpu_parity.py tests/test_gpu_config3.py tests/test_adaptive.py tests/test_gpu_closed_forms.py tests/test_gpu_tail_scores.py -x -q -s > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"
grep -h "float routines\|logf:" $O/pytest.txt | cut -c1-600
