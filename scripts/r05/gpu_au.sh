#!/bin/bash
# round 5, call AU: the margin scale of a pair's FIRST lists (CVO_HIP_FIRST_SCALE; default: 1 for single handles, 1.75 for batches of a few pairs on >= 4 workgroups each) with the final kernel
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_au; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt; : > $O/latency.txt
for rep in 1 2; do for v in default 1 1.4 1.75 2.2; do
  if [ $v = default ]; then unset CVO_HIP_FIRST_SCALE; else export CVO_HIP_FIRST_SCALE=$v; fi
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config5 > $O/bench.json 2> $O/bench.err || exit 1
  python - $v <<'PY' | tee -a $O/latency.txt
import json, sys; d=json.loads(open('gpurun_out/r5_au/bench.json').read().strip().splitlines()[-1]); l=d['latency']; print('first_scale', sys.argv[1], {k: round(l[k],3) for k in ('single_pair_align_ms','tracker_frame_from_images_ms','tracker_frame_next_frame_staged_ms','lc_batch_align_ms')})
PY
done; done
