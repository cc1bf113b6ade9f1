#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_pcd2; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_pcd.py tests/test_gpu_replay.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2 3; do timeout -k 10 200 python scripts/gpu_r4_tracker2.py 2>&1 | grep "queued score block on" | cut -c1-175; done | tee $O/pieces.txt
