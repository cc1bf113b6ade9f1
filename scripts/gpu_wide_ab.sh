#!/bin/bash
# plane-layout launches with three waves per SIMD (the library's default) against two (CVO_HIP_WIDE=0), same library, interleaved
for rep in 1 2 3; do for w in 0 1; do
  v=$(CVO_HIP_WIDE=$w timeout -k 10 300 python bench.py --shape eth3d --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep eth3d steps 24 CVO_HIP_WIDE=$w: $v"
done; done
for w in 0 1; do
  v=$(CVO_HIP_WIDE=$w timeout -k 10 300 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "tum steps 64 CVO_HIP_WIDE=$w (not a plane-layout launch: no difference expected): $v"
done
