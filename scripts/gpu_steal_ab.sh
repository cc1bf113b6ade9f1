#!/bin/bash
# (historical: needs the -DCVO_STEAL build of scripts/experiments/r03_block_stealing.patch as tmp_libs/libcvo_hip_steal.so)
# blocks of the steady candidate walk taken dynamically by the waves (-DCVO_STEAL build) against the static serpentine deal: parity subset first, then the A/B
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_steal.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adoption.py tests/test_gpu_config5.py tests/test_gpu_pair_order.py -x -q 2>&1 | tail -3
run() { v=$(CVO_BENCH_PHASES=1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$1.so timeout -k 10 300 python bench.py --shape $2 --steps $3 --warmup $4 --no-cpu-baseline --no-latency-probe 2>gpurun_out/steal.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))"); echo "rep $rep $2 steps $3 $1: $v | $(grep 'phase us' gpurun_out/steal.err | sed 's/.*launch): //' | cut -c1-200)"; }
for rep in 1 2 3; do
  run base tum 256 16; run steal tum 256 16
  run base tum 20 5; run steal tum 20 5
  run base eth3d 24 4; run steal eth3d 24 4
done
