#!/bin/bash
# round 5, call AL: latency object only, new against prev, five rounds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_al; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
: > $O/latency.txt
for rep in 1 2 3 4 5; do for v in new prev; do
  if [ $v = prev ]; then export CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so; else unset CVO_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config5 > $O/bench_$v.json 2> $O/bench_$v.err || exit 1
  python - $v <<'PY' | tee -a $O/latency.txt
import json, sys; d=json.loads(open(f'gpurun_out/r5_al/bench_{sys.argv[1]}.json').read().strip().splitlines()[-1]); l=d['latency']; print(sys.argv[1], {k: round(l[k],3) for k in ('single_pair_align_ms','tracker_frame_from_images_ms','tracker_frame_next_frame_staged_ms','lc_batch_align_ms')})
PY
done; done
unset CVO_HIP_LIB
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" -- "even_where_scarce" "prev CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_prev.so" | cut -c1-100
