#!/bin/bash
# config 5: the four workgroups of a pair on one XCD (CVO_HIP_COLOCATE, needs pair slots in multiples of 8) against consecutive blocks
for rep in 1 2; do for cfg in "40 6 1" "32 8 0" "32 8 1" "64 4 0" "64 4 1" "32 7 1" "32 6 1"; do
  read -r mw st co <<< "$cfg"
  r=$(CVO_HIP_COLOCATE=$co timeout -k 10 300 python bench.py --shape eth3d --workgroups 4 --max-workgroups $mw --streams $st --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],1))")
  echo "rep $rep max-workgroups $mw streams $st colocate $co: $r"
done; done
