#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_predict; mkdir -p $O
for p in 0 0.7; do CVO_HIP_PREDICT=$p CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-latency-probe 2>&1 >/dev/null | grep "culls by iteration" | grep -o "around.*"; done
