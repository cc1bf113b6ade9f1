#!/bin/bash
# config 5 on the three-wave build: workgroups per pair x launch share x launches side by side, once more (safe geometries only: never more cooperating workgroups than CUs)
for cfg in "4 64 4" "4 128 2" "4 32 8" "8 64 4" "8 128 2" "3 48 5" "3 63 4" "6 48 5" "5 60 4" "4 64 4"; do
  read -r g mw st <<< "$cfg"
  r=$(timeout -k 10 300 python bench.py --shape eth3d --workgroups $g --max-workgroups $mw --streams $st --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],1))")
  echo "G $g max-workgroups $mw streams $st: $r"
done
