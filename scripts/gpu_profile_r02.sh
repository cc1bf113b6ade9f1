#!/bin/bash
# round-2 profiles at HEAD: kernel-trace stats of the bench command (default and driver-style), PMC passes unloaded and under load
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02; rm -rf $O; mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 bench.py --no-cpu-baseline --no-latency-probe > $O/trace_default.json 2> $O/trace_default.err; echo "trace default rc=$?"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_driver -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-probe > $O/trace_driver.json 2> $O/trace_driver.err; echo "trace driver rc=$?"
# the counter passes measure the work of a launch: adoption off (with one launch alone on the device the idle CUs would all help and add their list rebuilds)
BENCH_ARGS="--steps 3 --warmup 1 --streams 1 --no-adoption" bash scripts/pmc_run.sh $O/pmc_alone sq1 sq2 sq3 sq4 sq5 fetch write
BENCH_ARGS="--steps 16 --warmup 8 --streams 8 --no-adoption" bash scripts/pmc_run.sh $O/pmc_load sq1
timeout -k 5 120 scripts/micro/valu_rate > $O/valu_issue_microbench.txt 2>&1; echo "microbench rc=$?"
find $O -name "*kernel_stats.csv" | head; du -sh $O
