#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_runpair; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py tests/test_gpu_parity.py tests/test_gpu_config5.py tests/test_gpu_multi.py tests/test_gpu_handover.py -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.txt)"
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "tum 20 5" "tum 256 32" "eth3d 24 4" -- "head"
