#!/bin/bash
# round 5, call J: the hand-over loop with and without the positions staged through LDS, from the ring and from registered memory (one box, interleaved)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_j; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
for rep in 1 2 3; do for st in 0 1; do for steps in "20 5" "256 32"; do
  set -- $steps
  CVO_HIP_PACK_STAGE=$st CVO_BENCH_NO_DISTINCT_LOOP=1 timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-config5 2> $O/err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); u = d['with_host_upload']
print('rep $rep stage $st steps $1: value', round(d['value']), 'upload', round(u['value']), round(u['value']/d['value'], 3), 'registered', round(u['from_registered_memory']['value']), round(u['from_registered_memory']['value']/d['value'], 3), 'score', round(d['with_score_block']['value']/d['value'], 3))" | tee -a $O/upload_ab.txt
done; done; done
