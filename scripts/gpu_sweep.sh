#!/bin/bash
# throughput of bench.py for several (workgroups per pair, steps in flight[, GPU_MAX_HW_QUEUES])
CFGS=${CFGS:-"4 2 4;2 4 4;1 8 4;1 8 8;1 16 16;2 8 8;2 6 8"}
IFS=';' read -ra LIST <<< "$CFGS"
for cfg in "${LIST[@]}"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$3 timeout -k 10 150 python bench.py --no-cpu-baseline --steps ${STEPS:-64} --warmup 16 --workgroups $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('G=$1 streams=$2 hwq=$3', round(d['value'],1), 'align/s', 'ms/step', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"
done
