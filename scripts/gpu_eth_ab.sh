#!/bin/bash
# A/B of library builds on the BASELINE config 5 shape (development aid): usage gpu_eth_ab.sh "<label>|<env>" ...
: > gpurun_out/eth_ab.txt
for rep in 1 2; do for cfg in "$@"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  v=$(env $envs CVO_BENCH_PHASES=1 timeout -k 10 400 python bench.py --shape eth3d --no-cpu-baseline --no-latency-probe 2> gpurun_out/eth_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],1))")
  ph=$(grep "phase us" gpurun_out/eth_ab.err | sed 's/.*launch): //')
  echo "$label: $v $ph" | tee -a gpurun_out/eth_ab.txt
done; done
