#!/bin/bash
# adoption's iteration cutoff (CVO_HIP_ADOPT_KMAX): same lease, interleaved; 0 = adoption armed but nobody ever helps (its fixed cost)
OUT=${1:-gpurun_out/adopt_kmax.txt}; : > $OUT
for rep in 1 2 3; do for cfg in "20 5" "256 32"; do for k in off 0 12 20 30 1000; do
  set -- $cfg
  if [ $k = off ]; then mode=--no-adoption; kk=20; else mode=--adoption; kk=$k; fi
  v=$(CVO_HIP_ADOPT_KMAX=$kk timeout -k 10 200 python bench.py --steps $1 --warmup $2 $mode --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))") || { echo "run failed ($k $cfg)" | tee -a $OUT; exit 1; }
  echo "rep $rep steps $1 kmax $k: $v" | tee -a $OUT
done; done; done
