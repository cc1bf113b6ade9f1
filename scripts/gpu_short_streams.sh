#!/bin/bash
# driver-style short run: steps in flight (development aid)
for rep in 1 2; do for st in 8 10 12 16; do
  v=$(GPU_MAX_HW_QUEUES=$st timeout -k 10 200 python bench.py --streams $st --steps 20 --warmup 5 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2))")
  echo "streams $st: $v"
done; done
