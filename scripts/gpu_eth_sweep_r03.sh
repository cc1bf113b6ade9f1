#!/bin/bash
# config 5 shape: workgroups per pair x launch share x launches side by side (pairs pulled from the in-kernel queue), one lease
for cfg in "4 40 6" "4 32 8" "4 48 5" "6 42 6" "6 36 7" "8 40 6" "8 32 8" "3 42 6" "4 40 6"; do
  read -r g mw st <<< "$cfg"
  r=$(timeout -k 10 300 python bench.py --shape eth3d --workgroups $g --max-workgroups $mw --streams $st --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],1))")
  echo "G $g max-workgroups $mw streams $st: $r"
done
