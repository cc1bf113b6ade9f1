#!/bin/bash
# TUM shape: launches with fewer slots than pairs (in-kernel queue, densest first) against a slot per pair; experiment orders 10 / 11
run() { # label steps warm env... -- args
  label=$1; steps=$2; warm=$3; shift 3
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  v=$(env GPU_MAX_HW_QUEUES=16 "${envs[@]}" timeout -k 10 300 python bench.py --steps $steps --warmup $warm --no-cpu-baseline --no-latency-probe "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['kernel_ms'])")
  echo "rep $rep steps $steps $label: $v"
}
for rep in 1 2; do for cfg in "256 32" "20 5"; do read -r steps warm <<< "$cfg"
  run "slot per pair x 8" $steps $warm X=1 --
  run "order 10 (class, lightest first)" $steps $warm CVO_HIP_ORDER_PAIRS=10 --
  run "order 11 (classes reversed)" $steps $warm CVO_HIP_ORDER_PAIRS=11 --
  run "32 slots x 8" $steps $warm X=1 -- --max-workgroups 32 --streams 8
  run "32 slots x 12" $steps $warm X=1 -- --max-workgroups 32 --streams 12
  run "48 slots x 8" $steps $warm X=1 -- --max-workgroups 48 --streams 8
  run "16 slots x 16" $steps $warm X=1 -- --max-workgroups 16 --streams 16
done; done
