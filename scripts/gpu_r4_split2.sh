#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_split; mkdir -p $O
for k in 2; do
  CVO_HIP_SPLIT_ROWS=$k CVO_HIP_SPLIT_MIN_G=2 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 | tee $O/pytest2_k$k.txt
  grep -q passed $O/pytest2_k$k.txt && ! grep -q failed $O/pytest2_k$k.txt || exit 1
done
for rep in 1 2; do for k in 0 2 4; do echo "== CVO_HIP_SPLIT_ROWS=$k"; CVO_HIP_SPLIT_ROWS=$k WGS=8 PAIR=0,5,9,17 timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids | cut -c1-420; done; done | tee $O/phases2.txt
