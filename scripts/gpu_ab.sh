#!/bin/bash
# A/B of library builds with the phase breakdown (development aid): usage gpu_ab.sh "<label>|<env>" ...
: > gpurun_out/ab.txt
for rep in 1 2; do for cfg in "$@"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  env $envs CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --pairs 64 --streams 8 --steps 64 --warmup 8 --no-cpu-baseline --no-latency-probe > gpurun_out/ab.json 2> gpurun_out/ab.err
  v=$(python -c "import sys,json; d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), round(d['config']['single_step_ms_unpipelined'],2))")
  ph=$(grep "phase us" gpurun_out/ab.err | sed 's/.*launch): //')
  echo "$label: $v $ph" | tee -a gpurun_out/ab.txt
done; done
