#!/bin/bash
# round-2 profiles of the BASELINE config 5 shape (736x456, ~9.3 k points): kernel stats + PMC
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02_eth; rm -rf $O; mkdir -p $O
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --shape eth3d --no-cpu-baseline --no-latency-probe > $O/trace.json 2> $O/trace.err; echo "trace rc=$?"
BENCH_ARGS="--shape eth3d --steps 2 --warmup 1 --streams 1 --max-workgroups 256" bash scripts/pmc_run.sh $O/pmc sq1 sq2 sq4 fetch write
find $O -name "*kernel_stats.csv"; tail -c 300 $O/trace.json
