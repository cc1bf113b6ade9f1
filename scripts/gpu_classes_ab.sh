#!/bin/bash
# the pair order rule with K density classes instead of 8 (CVO_HIP_ORDER_PAIRS = 20 + K): position p <- chunk p % K of the ranking
for rep in 1 2; do for k in 8 2 4 16 32; do
  v=$(CVO_HIP_ORDER_PAIRS=$((20 + k)) timeout -k 10 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],2))")
  echo "rep $rep steps 256 K=$k: $v"
done; done
