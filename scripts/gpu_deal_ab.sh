#!/bin/bash
# blocks dealt to the waves by cost (lpt) against the serpentine deal (serp), two and three waves per SIMD, same lease
run() { # lib shape steps warm env
  v=$(env $5 CVO_BENCH_PHASES=1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$1.so timeout -k 10 300 python bench.py --shape $2 --steps $3 --warmup $4 --no-cpu-baseline --no-latency-probe 2>gpurun_out/deal.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['parity']['max_rot_err_rad'] if d.get('parity') else '-')")
  echo "rep $rep $2 steps $3 $1 [$5]: $v | $(grep 'phase us' gpurun_out/deal.err | sed 's/.*launch): //' | cut -c1-210)"
}
for rep in 1 2 3; do
  run serp tum 256 16 X=1; run lpt tum 256 16 X=1
  run serp tum 20 5 X=1; run lpt tum 20 5 X=1
  run serp eth3d 24 4 X=1; run lpt eth3d 24 4 X=1
  run serp eth3d 24 4 CVO_HIP_WIDE=0; run lpt eth3d 24 4 CVO_HIP_WIDE=0
done
