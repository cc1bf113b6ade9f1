#!/bin/bash
# driver-style short run and a longer one, value / kernel ms / parity-free (development aid)
run() { echo "$1: $(env $2 timeout -k 10 300 python bench.py --pairs 64 --streams 8 --steps $3 --warmup $4 --no-cpu-baseline --no-latency-probe 2>gpurun_out/qb.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), d['config']['iterations_mean'], round(d['config']['single_step_ms_unpipelined'],2))" || tail -3 gpurun_out/qb.err)" | tee -a gpurun_out/qb.txt; }
: > gpurun_out/qb.txt
for rep in 1 2; do
run "table 20 steps" "A=1" 20 5
run "table 128 steps" "A=1" 128 16
run "no table 20 steps" "CVO_HIP_NO_TABLE=1" 20 5
run "no table 128 steps" "CVO_HIP_NO_TABLE=1" 128 16
done
