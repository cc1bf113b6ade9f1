#!/bin/bash
# BASELINE config 5 shape (736x456, ~9.3 k points per cloud): workgroups per pair x steps in flight, with the phase breakdown
: > gpurun_out/eth.txt
for cfg in "$@"; do
  set -- $cfg
  env $3 CVO_BENCH_PHASES=1 timeout -k 10 400 python bench.py --shape eth3d --workgroups $1 --streams $2 --pairs ${4:-64} --steps ${5:-8} --max-workgroups ${6:-0} --warmup 2 --no-cpu-baseline --no-latency-probe > gpurun_out/eth_ab.json 2> gpurun_out/eth_ab.err
  v=$(python -c "import sys,json; d=json.loads(open('gpurun_out/eth_ab.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), round(d['config']['single_step_ms_unpipelined'],2), d['config']['iterations_mean'])")
  ph=$(grep "phase us" gpurun_out/eth_ab.err | sed 's/.*launch): //')
  echo "G=$1 streams=$2 pairs=${4:-64} maxwg=${6:-0} $3: $v $ph" | tee -a gpurun_out/eth.txt
done
