#!/bin/bash
# round 5, call N: the steady walk fetches the next step's entries only when there is a next step (against: always, and waits for them)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_n; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py tests/test_gpu_adoption.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -eq 124 ]; then exit 1; fi
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" "eth3d 16 4" -- "always CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_pfall.so" "needed" | cut -c1-300
