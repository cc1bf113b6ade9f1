#!/bin/bash
# second pass of the CVO_HIP_SKIN sweep: larger margins
run() { label=$1; e=$2; shift 2
  v=$(env $e CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --shape $1 --steps $2 --warmup $3 --no-cpu-baseline --no-latency-probe 2>gpurun_out/skin.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep $1 steps $2 [$label]: $v | $(grep 'phase us' gpurun_out/skin.err | sed 's/.*launch): //' | sed "s/'cand_reduce.*'lists_cull'/ cull/" | cut -c1-200)"
}
for rep in 1 2; do
  for sk in 0.25 0.35 0.40 0.45 0.50 0.60; do run "skin $sk" CVO_HIP_SKIN=$sk tum 256 16; done
  for sk in 0.25 0.35 0.45 0.60; do run "skin $sk" CVO_HIP_SKIN=$sk tum 20 5; done
  for sk in 0.25 0.35 0.40 0.50; do run "skin $sk" CVO_HIP_SKIN=$sk eth3d 24 4; done
done
