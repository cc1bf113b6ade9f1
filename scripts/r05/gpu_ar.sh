#!/bin/bash
# round 5, call AR: the members' exchange polls back to back (clock read every 64th poll, no s_sleep) against clock + s_sleep per poll
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ar; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py tests/test_gpu_config5.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
: > $O/latency.txt
for rep in 1 2 3 4; do for v in new prev; do
  if [ $v = prev ]; then export CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so; else unset CVO_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config5 > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python - $v <<'PY' | tee -a $O/latency.txt
import json, sys; d=json.loads(open(f'gpurun_out/r5_ar/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1]); l=d['latency']; print(sys.argv[1], {k: round(l[k],3) for k in ('single_pair_align_ms','tracker_frame_from_images_ms','tracker_frame_next_frame_staged_ms','lc_batch_align_ms')})
PY
done; done
unset CVO_HIP_LIB
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "eth3d 16 4" -- "tight" "prev CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-70
