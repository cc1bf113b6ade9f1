#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_lsbal; mkdir -p $O
for rep in 1 2; do for b in 0 150 200 250 300; do
  CVO_HIP_LS_BALANCE=$b WGS=8 PAIR=0,5,9,17,22,31,40,55 timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids | python -c "
import sys,re
w=[]
for l in sys.stdin:
    m=re.search(r'wall ([0-9.]+) ms',l)
    if m: w.append(float(m.group(1)))
print('LS_BALANCE=$b: sum of 8 pairs %.3f ms; per pair %s' % (sum(w), ' '.join('%.2f'%x for x in w)))"
  CVO_HIP_LS_BALANCE=$b timeout -k 10 200 python scripts/gpu_r4_tracker2.py 2>&1 | grep "queued score block on" | cut -c25-140; done; done | tee $O/ab2.txt
