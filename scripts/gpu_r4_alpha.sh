#!/bin/bash
# round 4: depth-proportional list margin (DevParams::skin_alpha): parity with the margin on, then skin x alpha on the driver's command and the default run, one lease
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_alpha; mkdir -p $O; hostname > $O/lease.txt
CVO_HIP_SKIN_ALPHA=0.02 timeout -k 10 400 python -m pytest tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py -x -q > $O/pytest_alpha.txt 2>&1; echo "parity with alpha 0.02 rc=$? $(tail -1 $O/pytest_alpha.txt)"
bash scripts/gpu_ab_env.sh $O/sweep.txt 1 "tum 20 5" "tum 256 32" -- "base" "a005 CVO_HIP_SKIN_ALPHA=0.005" "a01 CVO_HIP_SKIN_ALPHA=0.01" "a02 CVO_HIP_SKIN_ALPHA=0.02" "a03 CVO_HIP_SKIN_ALPHA=0.03" \
  "s25a01 CVO_HIP_SKIN=0.25 CVO_HIP_SKIN_ALPHA=0.01" "s25a02 CVO_HIP_SKIN=0.25 CVO_HIP_SKIN_ALPHA=0.02" "s25a03 CVO_HIP_SKIN=0.25 CVO_HIP_SKIN_ALPHA=0.03" "s15a02 CVO_HIP_SKIN=0.15 CVO_HIP_SKIN_ALPHA=0.02" "s15a04 CVO_HIP_SKIN=0.15 CVO_HIP_SKIN_ALPHA=0.04" "base2"
