#!/bin/bash
# round-5 profiles at HEAD (one lease): kernel-trace stats of the bench command (default and driver-style) and of the config-5 shape, PMC passes of one
# launch alone (adoption off: the work of a launch; and on), PMC under load, the config-5 PMC passes.   usage: gpu_profile_r05.sh [tum|eth|all]
WHAT=${1:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r05; mkdir -p $O
hostname > $O/lease.txt; date -u +%FT%TZ >> $O/lease.txt
if [ $WHAT = tum ] || [ $WHAT = all ]; then
  rm -rf $O/trace_default $O/trace_driver $O/pmc_alone $O/pmc_adopt $O/pmc_load
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 bench.py --no-cpu-baseline --no-latency-probe > $O/trace_default.json 2> $O/trace_default.err; echo "trace default rc=$?"
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_driver -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-probe > $O/trace_driver.json 2> $O/trace_driver.err; echo "trace driver rc=$?"
  BENCH_ARGS="--steps 3 --warmup 1 --streams 1 --no-adoption" bash scripts/pmc_run.sh $O/pmc_alone sq1 sq2 sq3 sq4 sq5 fetch write
  BENCH_ARGS="--steps 3 --warmup 1 --streams 1 --adoption" bash scripts/pmc_run.sh $O/pmc_adopt sq1 sq2
  BENCH_ARGS="--steps 16 --warmup 8 --streams 8 --adoption" bash scripts/pmc_run.sh $O/pmc_load sq1 sq3
fi
if [ $WHAT = eth ] || [ $WHAT = all ]; then
  rm -rf $O/eth_trace $O/eth_pmc
  timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eth_trace -- python3 bench.py --shape eth3d --no-cpu-baseline --no-latency-probe > $O/eth_trace.json 2> $O/eth_trace.err; echo "trace eth rc=$?"
  BENCH_ARGS="--shape eth3d --steps 2 --warmup 1 --streams 1 --max-workgroups 256" bash scripts/pmc_run.sh $O/eth_pmc sq1 sq2 sq4 fetch write
fi
if [ $WHAT = ethmask ] || [ $WHAT = all ]; then   # the experiment build that masks the entry loads of rows that have ended (scripts/experiments/r05_masked_entry_loads.patch): its traffic
  rm -rf $O/eth_pmc_masked
  if [ -f tmp_libs/libcvo_hip_mdef.so ]; then
    export CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_mdef.so
    BENCH_ARGS="--shape eth3d --steps 2 --warmup 1 --streams 1 --max-workgroups 256" bash scripts/pmc_run.sh $O/eth_pmc_masked fetch write
    unset CVO_HIP_LIB
  fi
fi
find $O -name "*kernel_stats.csv" | head; du -sh $O
