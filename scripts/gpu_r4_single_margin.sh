#!/bin/bash
# one pair alone on eight workgroups: do wider list margins (fewer culls, longer walks) pay when the walks are short?  env knobs of the list margin on the single-pair probe
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_single_margin; mkdir -p $O; hostname > $O/lease.txt
for v in "default" "CVO_HIP_FIRST_SCALE=1.5" "CVO_HIP_FIRST_SCALE=2" "CVO_HIP_FIRST_SCALE=2.5" "CVO_HIP_SKIN=0.08 CVO_HIP_SKIN_ALPHA=0.02" "CVO_HIP_SKIN=0.1 CVO_HIP_SKIN_ALPHA=0.03" "CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.02" "CVO_HIP_PREDICT=1.0" "default"; do
  echo "== $v"; envs=$v; [ "$v" = default ] && envs="X=1"
  env $envs WGS=8 PAIR=0,5,9,17 timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
done | tee $O/sweep.txt
