#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_latency; mkdir -p $O; hostname > $O/lease.txt
for c in 0 1; do for p in 0 5; do echo "CVO_HIP_COLOCATE=$c pair $p"; CVO_HIP_COLOCATE=$c PAIR=$p WGS=0,16,8,6,4 timeout -k 10 200 python scripts/gpu_latency.py 2>&1 | grep -v amdgpu.ids; done; done | tee $O/colocate_single.txt
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tail_scores.py tests/test_gpu_replay.py -x -q 2>&1 | tail -2
