#!/bin/bash
# driver-style short run (20 steps, 5 warm-up): steps in flight x hardware queues (development aid)
OUT=gpurun_out/short_sweep.txt; : > $OUT
for rep in 1 2; do for cfg in "8 8" "12 12" "16 16" "20 20" "24 24" "8 16" "8 20" "16 20"; do
  set -- $cfg
  v=$(GPU_MAX_HW_QUEUES=$1 timeout -k 10 200 python bench.py --streams $2 --steps 20 --warmup 5 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), round(d['roofline']['launches_side_by_side'],2))")
  echo "queues $1 streams $2: $v" | tee -a $OUT
done; done
