#!/bin/bash
# K lanes per row in the steady candidate walk of cooperating workgroups (cand_steady_split): parity, then the single-pair probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_split; mkdir -p $O; hostname > $O/lease.txt
for k in 4 2; do
  CVO_HIP_SPLIT_ROWS=$k CVO_HIP_SPLIT_MIN_G=2 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tail_scores.py tests/test_gpu_replay.py tests/test_gpu_pcd.py -x -q 2>&1 | tail -4 | tee $O/pytest_k$k.txt
  grep -q passed $O/pytest_k$k.txt && ! grep -q failed $O/pytest_k$k.txt || exit 1
done
for rep in 1 2; do for k in 0 2 4; do echo "== CVO_HIP_SPLIT_ROWS=$k"; CVO_HIP_SPLIT_ROWS=$k WGS=8,16,4 PAIR=0,5,9 timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids | cut -c1-420; done; done | tee $O/phases.txt
