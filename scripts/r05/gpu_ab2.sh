#!/bin/bash
# round 5, call AB: the new lists-stay-valid test on the fixed library (must pass) and on the library of the commit before (expected to fail on the cull count for the sizes whose points outnumber the pre-loaded ones)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ab2; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "between_the_layouts" tests/test_gpu_config5.py -q > $O/pytest_new.txt 2>&1; echo "new rc=$? $(tail -1 $O/pytest_new.txt)"
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -k "between_the_layouts" -q > $O/pytest_prev.txt 2>&1; echo "prev rc=$? $(tail -1 $O/pytest_prev.txt)"; grep -E "^FAILED|assert" $O/pytest_prev.txt | head -8
