#!/bin/bash
# round 5, call U: where the host's time per step goes in the bench loop
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_u; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
for s in "20 5" "256 32"; do set -- $s; CVO_BENCH_HOSTPROF=1 timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-config5 --no-latency-probe --parity-only > $O/b_$1.json 2> $O/b_$1.err; grep "host us" $O/b_$1.err; python -c "import json; d=json.loads(open('$O/b_$1.json').read().strip().splitlines()[-1]); print(d['value'])"; done
