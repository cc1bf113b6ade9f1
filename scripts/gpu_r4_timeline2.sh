#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_timeline2; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 200 python scripts/gpu_timeline.py --repeat 2 --out $O/on.json > $O/on.txt 2>&1; echo "on rc=$?"; grep -E "^repeat|idle" $O/on.txt | cut -c1-400
CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-latency-probe 2>&1 >/dev/null | grep -o "phase us.*"
