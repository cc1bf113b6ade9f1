#!/bin/bash
# kernel-trace stats of the three bench commands only (the PMC passes of gpu_profile_r02.sh / gpu_profile_eth.sh stay as they are)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02; mkdir -p $O; rm -rf $O/trace_default $O/trace_driver
E=gpurun_out/prof_r02_eth; mkdir -p $E; rm -rf $E/trace
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 bench.py --no-cpu-baseline --no-latency-probe > $O/trace_default.json 2> $O/trace_default.err; echo "trace default rc=$?"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_driver -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-probe > $O/trace_driver.json 2> $O/trace_driver.err; echo "trace driver rc=$?"
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $E/trace -- python3 bench.py --shape eth3d --no-cpu-baseline --no-latency-probe > $E/trace.json 2> $E/trace.err; echo "trace eth rc=$?"
