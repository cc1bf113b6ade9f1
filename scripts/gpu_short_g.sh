#!/bin/bash
# driver-style short run (20 steps, 5 warm-up) and the long run: workgroups per pair x steps in flight (development aid)
OUT=gpurun_out/short_g.txt; : > $OUT
for rep in 1 2; do for cfg in "1 8 20 5" "2 8 20 5" "2 4 20 5" "4 8 20 5" "4 4 20 5" "4 2 20 5" "1 8 128 8" "2 8 128 8" "4 4 128 8"; do
  set -- $cfg
  v=$(timeout -k 10 200 python bench.py --workgroups $1 --streams $2 --steps $3 --warmup $4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), round(d['roofline']['launches_side_by_side'],2), d['config']['workgroups_per_pair'])")
  echo "G $1 streams $2 steps $3: $v" | tee -a $OUT
done; done
