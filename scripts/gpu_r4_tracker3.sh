#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_tracker3; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
grep -q passed $O/pytest.txt && ! grep -q failed $O/pytest.txt || exit 1
timeout -k 10 200 python scripts/gpu_r4_tracker2.py 2>&1 | grep -v amdgpu.ids | tee $O/pieces.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 2> $O/bench_err.txt | tee $O/bench_driver.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['latency'])"
