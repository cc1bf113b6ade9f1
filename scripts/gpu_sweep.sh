#!/bin/bash
# throughput of bench.py for several (workgroups per pair, steps in flight)
for cfg in "4 2" "2 4" "1 4" "1 8" "2 8" "1 12"; do
  set -- $cfg
  timeout -k 10 150 python bench.py --no-cpu-baseline --steps 48 --warmup 12 --workgroups $1 --streams $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('G=$1 streams=$2', round(d['value'],1), 'align/s', 'ms/step', round(d['ms_per_step'],2), 'kernel_ms', round(d['roofline']['kernel_ms'],2))"
done
