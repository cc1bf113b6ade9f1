#!/bin/bash
# A/B of several builds of the library (CVO_HIP_LIB) on one lease, interleaved: usage gpu_libs_ab.sh REPS "shape steps warmup" ... -- lib.so ...
REPS=$1; shift
CFGS=(); while [ "$1" != "--" ]; do CFGS+=("$1"); shift; done; shift
LIBS=("$@")
for rep in $(seq 1 $REPS); do for cfg in "${CFGS[@]}"; do for lib in "${LIBS[@]}"; do
  read -r shape steps warm <<< "$cfg"
  v=$(CVO_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --shape $shape --steps $steps --warmup $warm --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep $shape steps $steps $(basename $lib): $v"
done; done; done
