#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_single; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 200 python scripts/gpu_r4_single_phases.py 2>&1 | grep -v amdgpu.ids | tee $O/phases.txt
