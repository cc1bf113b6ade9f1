#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_single_margin; mkdir -p $O
for v in "default" "CVO_HIP_FIRST_SCALE=2" "CVO_HIP_FIRST_SCALE=1.75" "CVO_HIP_FIRST_SCALE=2 CVO_HIP_SKIN=0.08 CVO_HIP_SKIN_ALPHA=0.02" "CVO_HIP_FIRST_SCALE=2 CVO_HIP_PREDICT=1.0" "CVO_HIP_FIRST_SCALE=1.5 CVO_HIP_SKIN=0.08 CVO_HIP_SKIN_ALPHA=0.02" "CVO_HIP_SKIN=0.08 CVO_HIP_SKIN_ALPHA=0.02" "default"; do
  envs=$v; [ "$v" = default ] && envs="X=1"
  env $envs WGS=8 PAIR=0,5,9,17,22,31,40,55 timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys,re
w=[];rb=[]
for l in sys.stdin:
    m=re.search(r'wall ([0-9.]+) ms.*rebuilds (\d+)',l)
    if m: w.append(float(m.group(1))); rb.append(int(m.group(2)))
print('$v: sum of 8 pairs %.3f ms; per pair %s; rebuilds %s' % (sum(w), ' '.join('%.2f'%x for x in w), rb))"
done | tee $O/sweep2.txt
