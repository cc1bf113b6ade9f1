#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_tail; mkdir -p $O; hostname > $O/lease.txt
CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; grep "score block in the tail" $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['with_score_block']['value']/d['value'], d['with_host_upload']['value']/d['value'])"
