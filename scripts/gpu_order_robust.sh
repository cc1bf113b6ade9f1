#!/bin/bash
# does the order in which a launch's positions take the pairs matter on other inputs too?  three other sets of 64 pairs, index / strided / alternating / densest, default run
for off in 64 128 192; do for m in 0 5 4 2; do
  v=$(CVO_BENCH_PAIR_OFFSET=$off CVO_HIP_ORDER_PAIRS=$m timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['config']['iterations_mean'])")
  echo "pairs $off..$((off+63)) order mode $m: $v"
done; done
