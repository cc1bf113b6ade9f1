#!/bin/bash
# driver-style short run and the long run with and without adoption (CVO_HIP_ADOPT: finished workgroups help with running pairs)
OUT=gpurun_out/adopt.txt; : > $OUT
for rep in 1 2 3; do for cfg in "0 20 5" "1 20 5" "0 128 8" "1 128 8"; do
  set -- $cfg
  v=$(CVO_HIP_ADOPT=$1 timeout -k 10 200 python bench.py --steps $2 --warmup $3 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), d['config']['iterations_mean'])")
  echo "adopt $1 steps $2: $v" | tee -a $OUT
done; done
