#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_deal; mkdir -p $O; hostname > $O/lease.txt
for rep in 1 2; do
for v in default deal64; do
  if [ $v = default ]; then unset CVO_HIP_LIB; else export CVO_HIP_LIB=$GRAFT_REPO_ROOT/tmp_libs/libcvo_hip_$v.so; fi
  echo "== $v"; WGS=8,4 timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids
done; done | tee $O/phases.txt
export CVO_HIP_LIB=$GRAFT_REPO_ROOT/tmp_libs/libcvo_hip_deal64.so
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | tee $O/pytest_deal64.txt
