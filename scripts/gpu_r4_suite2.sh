#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_suite2; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.txt)"
for r in 1 2; do
CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$r.json 2> $O/bench$r.err; python -c "
import json; d=json.loads(open('$O/bench$r.json').read().strip().splitlines()[-1]); print('driver', round(d['value']), 'score', round(d['with_score_block']['value']/d['value'],3), 'upload', round(d['with_host_upload']['value']/d['value'],3), 'distinct', round(d['distinct_pairs']['value']/d['value'],3))"
done
