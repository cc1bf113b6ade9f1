#!/bin/bash
# steps in flight at the round's final state (XCD-aware pair order): default run and the driver's command
for rep in 1 2; do for st in 6 8 10 12; do
  for cfg in "256 32" "20 5"; do read -r steps warm <<< "$cfg"
  v=$(GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python bench.py --streams $st --steps $steps --warmup $warm --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
  echo "rep $rep streams $st steps $steps: $v"; done
done; done
