#!/bin/bash
# round 4: the next iteration's transform begun before this iteration's second stop test is done (DevParams::overlap_stop_test): parity, then A/B on one lease
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_overlap; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 500 python -m pytest tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py tests/test_gpu_parity.py tests/test_gpu_config5.py tests/test_gpu_closed_forms.py -x -q > $O/pytest.txt 2>&1; echo "parity rc=$? $(tail -1 $O/pytest.txt)"
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" -- "behind CVO_HIP_OVERLAP_STOP=0" "overlapped CVO_HIP_OVERLAP_STOP=1"
