#!/bin/bash
# config 5: list entries read non-temporally (as the 3 k-point shape does) against plain, at the round's final state
for rep in 1 2 3; do for lib in n0 ntall; do
  v=$(CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$lib.so timeout -k 10 300 python bench.py --shape eth3d --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep eth3d steps 24 $lib: $v"
done; done
