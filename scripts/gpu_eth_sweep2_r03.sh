#!/bin/bash
# config 5 at the round's final state (densest-first queue, co-located members): launch share x launches side by side
for cfg in "64 4" "128 2" "64 3" "256 1" "48 5" "96 2" "96 3" "64 5" "40 6" "64 4"; do
  read -r mw st <<< "$cfg"
  r=$(timeout -k 10 300 python bench.py --shape eth3d --workgroups 4 --max-workgroups $mw --streams $st --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],1))")
  echo "G 4 max-workgroups $mw streams $st: $r"
done
for cfg in "8 64 4" "8 128 2" "6 48 5" "2 64 4" "2 32 8"; do
  read -r g mw st <<< "$cfg"
  r=$(timeout -k 10 300 python bench.py --shape eth3d --workgroups $g --max-workgroups $mw --streams $st --steps 24 --warmup 4 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],1))")
  echo "G $g max-workgroups $mw streams $st: $r"
done
