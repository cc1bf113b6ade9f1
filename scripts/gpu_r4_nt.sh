#!/bin/bash
# round 4: when do the lists go stale (culls by iteration), and short lists read with plain instead of non-temporal loads (CVO_HIP_NT_MIN entries)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_nt; mkdir -p $O; hostname > $O/lease.txt
CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 64 --warmup 16 --no-cpu-baseline --no-latency-probe 2>&1 >/dev/null | grep "culls by iteration" | tee $O/culls_by_iteration.txt
bash scripts/gpu_ab_env.sh $O/nt.txt 2 "tum 20 5" "tum 256 32" -- "base" "nt20k CVO_HIP_NT_MIN=20000" "nt60k CVO_HIP_NT_MIN=60000" "nt150k CVO_HIP_NT_MIN=150000" "ntall CVO_HIP_NT_MIN=100000000"
