#!/bin/bash
# round 5, call C: the default build (lean line search, every store and list load of the walks through buffer resources) through the parity suites; A/B against the
# branchy stores and with the degree-7 exp (coefficients in scalar registers this time)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_c; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
run() { local name=$1; shift; "$@" > $O/$name.txt 2>&1; local rc=$?; echo "$name rc=$rc $(tail -2 $O/$name.txt | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi; }
run pairs timeout -k 10 600 python -m pytest tests/test_gpu_pair_values.py -q -s
run parity timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py tests/test_gpu_adoption.py tests/test_gpu_config5.py -x -q
export CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_e7b.so
run parity_e7 timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py -x -q
unset CVO_HIP_LIB
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "tum 20 5" "tum 256 32" "eth3d 12 4" -- "branchy CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_branchy.so" "default" "e7b CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_e7b.so"
