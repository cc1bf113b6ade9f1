#!/bin/bash
# round 5, call B: libm KAT again; parity of the default build (lean line search + record stores through a buffer resource); A/B of the record stores and the degree-7 exp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_b; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
run() { local name=$1; shift; "$@" > $O/$name.txt 2>&1; local rc=$?; echo "$name rc=$rc $(tail -2 $O/$name.txt | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi; }
run pairs timeout -k 10 600 python -m pytest tests/test_gpu_pair_values.py -q -s
run parity timeout -k 10 900 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py tests/test_gpu_adoption.py -x -q
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "tum 20 5" "tum 256 32" -- "ls CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_ls.so" "lsbuf CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_lsbuf.so" "lse7 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_lse7.so" "lsbufe7 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_lsbufe7.so"
