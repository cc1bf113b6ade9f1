#!/bin/bash
# round 5, call AK: every wave an equal share of the workgroup's nonzero records in the line search (against: the segment it compacted itself)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ak; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_ktrace.so timeout -k 10 300 python scripts/r05/probe_iter.py > $O/iter.txt 2>&1; grep -A8 "wgs 8" $O/iter.txt | cut -c1-150
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py tests/test_gpu_config5.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
: > $O/latency.txt
for rep in 1 2 3; do for v in new prev; do
  if [ $v = prev ]; then export CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so; else unset CVO_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python - $v <<'PY' | tee -a $O/latency.txt
import json, sys; d=json.loads(open(f'gpurun_out/r5_ak/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1]); l=d['latency']; print(sys.argv[1], round(d['value']), {k: round(l[k],3) for k in ('single_pair_align_ms','tracker_frame_from_images_ms','tracker_frame_next_frame_staged_ms','lc_batch_align_ms')})
PY
done; done
unset CVO_HIP_LIB
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" "eth3d 16 4" -- "even_shares" "prev CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-100
