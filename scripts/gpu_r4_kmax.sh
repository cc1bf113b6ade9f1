#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_kmax; mkdir -p $O; hostname > $O/lease.txt
bash scripts/gpu_ab_env.sh $O/sweep.txt 2 "tum 20 5" -- "k20" "k0 CVO_HIP_ADOPT_KMAX=0" "k10 CVO_HIP_ADOPT_KMAX=10" "k40 CVO_HIP_ADOPT_KMAX=40" "k1000 CVO_HIP_ADOPT_KMAX=1000"
