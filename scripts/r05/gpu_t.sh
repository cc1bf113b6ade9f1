#!/bin/bash
# round 5, call T: two 256-thread workgroups per CU (CVO_HIP_WGS_PER_CU=2: half the LDS each) against one of 512, with the round-5 kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_t; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "tum 20 5" "tum 256 32" -- "one_per_cu" "two_per_cu CVO_HIP_WGS_PER_CU=2" | cut -c1-330
CVO_HIP_WGS_PER_CU=2 timeout -k 10 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-config5 --no-latency-probe --parity-only > $O/bench_two.json 2> $O/bench_two.err; tail -c 600 $O/bench_two.json
