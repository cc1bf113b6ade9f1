#!/bin/bash
# round 5, call AA: the epilogue's fused transform in the plane layout too (config 5)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_aa; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_config5.py tests/test_gpu_parity.py tests/test_gpu_adoption.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "eth3d 16 4" -- "fused" "prev CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-330
