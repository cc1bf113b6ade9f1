#!/bin/bash
# (historical: CVO_HIP_ROTATE existed only in the experiment build of that call -- profiles/r03_class_shift_across_handles.txt; the knob is not in the tree)
# the density classes' shift over the XCD classes: 0 none, 1 every batch object its own (0..7), 2 the same shift (3) in all, 3 two groups of objects (0 / 4); with and without adoption
for rep in 1 2; do for r in 0 1 2 3; do for ad in --adoption --no-adoption; do
  v=$(CVO_HIP_ROTATE=$r CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 256 --warmup 16 $ad --no-cpu-baseline --no-latency-probe 2>gpurun_out/rot.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],2))")
  echo "rep $rep steps 256 rotate $r $ad: $v | $(grep 'phase us' gpurun_out/rot.err | sed 's/.*launch): //' | cut -c1-175)"
done; done; done
