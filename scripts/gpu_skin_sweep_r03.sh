#!/bin/bash
# list radius margin (CVO_HIP_SKIN) and re-sort policy (CVO_HIP_RESORT) with eight launches in flight: memory costs more under load than alone
run() { # label env -- shape steps warm
  label=$1; e=$2; shift 2
  v=$(env $e CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --shape $1 --steps $2 --warmup $3 --no-cpu-baseline --no-latency-probe 2>gpurun_out/skin.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep $1 steps $2 [$label]: $v | $(grep 'phase us' gpurun_out/skin.err | sed 's/.*launch): //' | cut -c1-230)"
}
for rep in 1 2; do
  for sk in 0.25 0.15 0.20 0.30 0.35; do run "skin $sk" CVO_HIP_SKIN=$sk tum 256 16; done
  run "resort always" CVO_HIP_RESORT=2 tum 256 16
  run "resort never" CVO_HIP_RESORT=0 tum 256 16
  for sk in 0.25 0.15 0.20 0.30; do run "skin $sk" CVO_HIP_SKIN=$sk eth3d 24 4; done
done
