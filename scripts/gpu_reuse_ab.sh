#!/bin/bash
# which batch object the bench loop reuses (oldest / whichever is done) x adoption x steps in flight, one lease
for rep in 1 2 3; do for cfg in "20 5" "256 32"; do for v in "--reuse oldest --no-adoption" "--reuse any --no-adoption" "--reuse oldest --adoption" "--reuse any --adoption" "--reuse any --adoption --streams 12" "--reuse any --no-adoption --streams 12"; do
  read -r steps warm <<< "$cfg"
  r=$(GPU_MAX_HW_QUEUES=16 timeout -k 10 300 python bench.py --steps $steps --warmup $warm $v --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), round(d['roofline']['launches_side_by_side'],2))")
  echo "rep $rep steps $steps [$v]: $r"
done; done; done
