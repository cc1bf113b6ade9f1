#!/bin/bash
# round 5, call AQ: adoption's iteration cutoff with the final kernel (joins are cheaper now: cull on all waves, lists rebuilt together, line search shared)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_aq; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
bash scripts/gpu_ab_env.sh $O/ab.txt 4 "tum 20 5" "tum 256 32" -- "kmax20" "kmax40 CVO_HIP_ADOPT_KMAX=40" "kmax80 CVO_HIP_ADOPT_KMAX=80" "kmax1000 CVO_HIP_ADOPT_KMAX=1000" | cut -c1-70
