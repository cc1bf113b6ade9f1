#!/bin/bash
# round 5, call AC: the twist-only part of the pose update on an idle wave beside the step polynomial (CVO_TWIST_AHEAD): parity, the epilogue's timeline, latencies and throughput against the commit before
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ac; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config3.py tests/test_gpu_closed_forms.py tests/test_gpu_adoption.py tests/test_gpu_config5.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_epi2.so timeout -k 10 300 python scripts/r05/probe_epi2.py > $O/epi2.txt 2>&1; echo rc=$?; cut -c1-330 $O/epi2.txt
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" -- "twist_ahead" "prev CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-100
: > $O/latency.txt
for rep in 1 2 3; do for v in new prev; do
  if [ $v = prev ]; then export CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so; else unset CVO_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python - $v <<'PY' | tee -a $O/latency.txt
import json, sys; d=json.loads(open(f'gpurun_out/r5_ac/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1]); l=d['latency']; print(sys.argv[1], round(d['value']), {k: round(l[k],3) for k in ('single_pair_align_ms','tracker_frame_from_images_ms','tracker_frame_next_frame_staged_ms','lc_batch_align_ms')})
PY
done; done
