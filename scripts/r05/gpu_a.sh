#!/bin/bash
# round 5, call A: device KAT of the pair arithmetic + libm; parity of the variant build; A/B of the lean line search and the degree-7 exp
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_a; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
run() { local name=$1; shift; "$@" > $O/$name.txt 2>&1; local rc=$?; echo "$name rc=$rc $(tail -2 $O/$name.txt | tr '\n' ' ')"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi; }
run pairs timeout -k 10 600 python -m pytest tests/test_gpu_pair_values.py -q
export CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_both.so
run parity_both timeout -k 10 700 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py -x -q
unset CVO_HIP_LIB
bash scripts/gpu_libs_ab.sh 2 "tum 20 5" "tum 256 32" -- tmp_libs/libcvo_hip_base.so tmp_libs/libcvo_hip_ls.so tmp_libs/libcvo_hip_e7.so tmp_libs/libcvo_hip_both.so 2>&1 | tee $O/ab.txt
