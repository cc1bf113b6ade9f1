#!/bin/bash
# round 4: the last candidate walk at an ell makes the next ell's lists on its way (DevParams::fuse_refine) against a filter pass of its own at the drop: parity, then A/B on one lease
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_fuse; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py tests/test_gpu_parity.py tests/test_gpu_config5.py tests/test_gpu_noise_envelope.py -x -q > $O/pytest.txt 2>&1; echo "parity rc=$? $(tail -1 $O/pytest.txt)"
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" -- "pass CVO_HIP_FUSE_REFINE=0" "fused CVO_HIP_FUSE_REFINE=1"
