#!/bin/bash
# three and four waves per SIMD (768 / 1024 threads per workgroup, 168 / 128 VGPRs) against two, with the device full
run() { # lib block shape steps warm
  v=$(CVO_HIP_BLOCK=$2 CVO_BENCH_PHASES=1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$1.so timeout -k 10 300 python bench.py --shape $3 --steps $4 --warmup $5 --no-cpu-baseline --no-latency-probe 2>gpurun_out/blk.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep $3 steps $4 $1 block $2: $v | $(grep 'phase us' gpurun_out/blk.err | sed 's/.*launch): //' | cut -c1-200)"
}
for rep in 1 2; do
  run gl 512 tum 256 16; run b768 768 tum 256 16; run b1024 1024 tum 256 16
  run gl 512 eth3d 24 4; run b768 768 eth3d 24 4; run b1024 1024 eth3d 24 4
done
