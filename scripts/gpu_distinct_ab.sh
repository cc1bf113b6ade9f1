#!/bin/bash
# every batch object its own 64 pairs (CVO_BENCH_DISTINCT=1: 512 different pairs in flight, not eight copies of one set): does the pair order rule still pay?
for rep in 1 2; do for o in 0 1; do
  v=$(CVO_BENCH_DISTINCT=1 CVO_HIP_ORDER_PAIRS=$o timeout -k 10 600 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],2))")
  echo "rep $rep distinct sets, steps 256, CVO_HIP_ORDER_PAIRS=$o: $v"
done; done
