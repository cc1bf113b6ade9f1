#!/bin/bash
# A/B of two builds of the library (CVO_HIP_LIB) on one lease, interleaved: usage gpu_lib_ab.sh REPS libA.so libB.so
REPS=$1; A=$2; B=$3
for rep in $(seq 1 $REPS); do for cfg in "20 5" "256 32"; do for lib in "$A" "$B"; do
  read -r steps warm <<< "$cfg"
  v=$(CVO_HIP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
  echo "rep $rep steps $steps $(basename $lib): $v"
done; done; done
