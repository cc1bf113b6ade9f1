#!/bin/bash
# the three-waves-per-SIMD build on the 3 k-point shape (float4 layout), after round 4 took most of its spills away: CVO_HIP_WIDE=2 against the default (two waves)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_w3; mkdir -p $O; hostname > $O/lease.txt
CVO_HIP_WIDE=2 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 | tee $O/pytest.txt
grep -q passed $O/pytest.txt && ! grep -q failed $O/pytest.txt || exit 1
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "tum 64 8" "tum 20 5" -- "two" "three CVO_HIP_WIDE=2"
