#!/bin/bash
# round 5, call L: large clouds resident as 12-byte points instead of three planes: parity (config 5 as benchmarked, plane-layout cases of the suites), then A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_l; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_parity.py tests/test_gpu_adoption.py tests/test_gpu_pair_order.py tests/test_gpu_tail_scores.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -eq 124 ]; then exit 1; fi
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "eth3d 16 4" -- "planes CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_planes.so" "y12" | cut -c1-330
