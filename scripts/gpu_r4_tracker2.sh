#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_tracker2; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 300 python -m pytest tests/test_gpu_pcd.py tests/test_gpu_tail_scores.py tests/test_gpu_replay.py -x -q 2>&1 | tail -15 | tee $O/pytest.txt
grep -q passed $O/pytest.txt && ! grep -q failed $O/pytest.txt || exit 1
timeout -k 10 200 python scripts/gpu_r4_tracker2.py 2>&1 | grep -v amdgpu.ids | tee $O/pieces.txt
CVO_HIP_SHARE_CLOUDS=0 timeout -k 10 200 python scripts/gpu_r4_tracker2.py 2>&1 | grep -v amdgpu.ids | tee $O/pieces_noshare.txt
