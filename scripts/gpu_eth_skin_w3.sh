#!/bin/bash
# config 5 on the three-wave build: list margin once more
for rep in 1 2; do for sk in 0.30 0.25 0.35 0.40; do
  v=$(CVO_HIP_SKIN=$sk timeout -k 10 300 python bench.py --shape eth3d --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep eth3d skin $sk: $v"
done; done
