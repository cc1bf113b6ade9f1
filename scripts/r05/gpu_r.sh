#!/bin/bash
# round 5, call R: the epilogue's transform with lane 0 off its critical path: parity, the timeline, the bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_r; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_epi2.so timeout -k 10 300 python scripts/r05/probe_epi2.py > $O/epi2.txt 2>&1; echo rc=$?; cat $O/epi2.txt
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" -- "new" "prev CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-100
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/bench_driver.json 2> $O/bench_driver.err; python - <<'PY'
import json; d=json.loads(open('gpurun_out/r5_r/bench_driver.json').read().strip().splitlines()[-1]); print(d['value'], d['latency'])
PY
