#!/bin/bash
# round 4: the hand-over's block copied to a device mirror by the copy engine (CVO_HIP_RING_MIRROR) against the kernel reading the pinned ring; with_host_upload / value of bench.py
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_upload; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 300 python -m pytest tests/test_gpu_handover.py tests/test_gpu_config3.py -x -q 2>&1 | tail -2
for r in 1 2 3; do for m in 0 1; do for st in "20 5" "256 32"; do set -- $st
CVO_HIP_RING_MIRROR=$m timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline > $O/b.json 2> $O/b.err; python -c "
import json; d=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); print('rep $r mirror $m steps $1: value', round(d['value']), 'upload', round(d['with_host_upload']['value']), round(d['with_host_upload']['value']/d['value'],3), 'score', round(d['with_score_block']['value']/d['value'],3))" | tee -a $O/ab.txt
done; done; done
