#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_first_scale; mkdir -p $O; hostname > $O/lease.txt
CVO_HIP_FIRST_SCALE=2.5 timeout -k 10 300 python -m pytest tests/test_gpu_config3.py tests/test_gpu_parity.py -x -q 2>&1 | tail -1
bash scripts/gpu_ab_env.sh $O/sweep.txt 1 "tum 20 5" "tum 256 32" -- "f1" "f15 CVO_HIP_FIRST_SCALE=1.5" "f2 CVO_HIP_FIRST_SCALE=2" "f3 CVO_HIP_FIRST_SCALE=3" "f4 CVO_HIP_FIRST_SCALE=4" "f1b"
CVO_HIP_FIRST_SCALE=2 CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-latency-probe 2>&1 >/dev/null | grep "culls by iteration" | sed -e 's/ 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0;/;/'
