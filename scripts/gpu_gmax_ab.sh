#!/bin/bash
# how many workgroups a pair may grow to by adoption (-DCVO_ADOPT_GMAX builds in tmp_libs): the 20-step command, where adoption matters
for rep in 1 2 3; do for lib in g4 g2 g3 g6; do
  v=$(CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$lib.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['parity']['max_rot_err_rad'] if d.get('parity') else '')")
  echo "rep $rep steps 20 $lib: $v"
done; done
