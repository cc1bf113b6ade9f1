#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_tracker; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 400 python -m pytest tests/test_gpu_tail_scores.py tests/test_gpu_parity.py tests/test_gpu_replay.py tests/test_gpu_pcd.py tests/test_cpp_mirror.py -x -q 2>&1 | tail -3
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['with_score_block']['value']/d['value'], d['with_host_upload']['value']/d['value']); print(d['latency'])"
