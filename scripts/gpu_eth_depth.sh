#!/bin/bash
# experiment: launches in flight x share of the device, 9.3 k-point clouds
for cfg in "4 64 24" "5 48 25" "6 40 24" "7 36 28" "8 32 32"; do set -- $cfg
  timeout -k 10 300 python bench.py --shape eth3d --workgroups 4 --streams $1 --max-workgroups $2 --steps $3 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $1 maxwg $2:', round(d['value']), 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'side by side', round(d['roofline']['launches_side_by_side'],2))"
done
