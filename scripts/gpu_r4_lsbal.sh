#!/bin/bash
# cooperating workgroups: the waves share the workgroup's nonzero records evenly in the line search (CVO_HIP_LS_BALANCE=1, default) against every wave its own segment (0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_lsbal; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tail_scores.py tests/test_gpu_replay.py tests/test_gpu_multi.py tests/test_gpu_config3.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
grep -q passed $O/pytest.txt && ! grep -q failed $O/pytest.txt || exit 1
for rep in 1 2; do for b in 0 1; do echo "== CVO_HIP_LS_BALANCE=$b"; CVO_HIP_LS_BALANCE=$b WGS=8,4 PAIR=0,5,9,17 timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids | cut -c1-330
  CVO_HIP_LS_BALANCE=$b timeout -k 10 200 python scripts/gpu_r4_tracker2.py 2>&1 | grep "queued score block on" | cut -c1-175; done; done | tee $O/ab.txt
