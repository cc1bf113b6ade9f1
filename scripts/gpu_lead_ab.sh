#!/bin/bash
# (historical: CVO_HIP_LEAD existed only with scripts/experiments/r03_lead_helpers.patch applied)
# lead helpers (CVO_HIP_LEAD = extra workgroups that start as helpers of the launch's densest pairs): same lease, interleaved
run() { v=$(CVO_HIP_LEAD=$1 timeout -k 10 300 python bench.py --steps $2 --warmup $3 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],2))"); echo "rep $rep steps $2 lead $1: $v"; }
for rep in 1 2 3; do
  for h in 0 2 4 8; do run $h 20 5; done
  for h in 0 2 4 8; do run $h 256 16; done
done
