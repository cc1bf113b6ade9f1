#!/bin/bash
# round 5, call AM: automatic workgroup count = at most ~384 rows per workgroup (was: at least): latencies on both shapes, full GPU suite
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_am; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
python scripts/r05/probe_tracker_g.py 2>&1 | grep "^workgroups" | head -5
for rep in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config5 > $O/bench_tum.json 2> $O/bench_tum.err || exit 1
  python - <<'PY' | tee -a $O/latency.txt
import json; d=json.loads(open('gpurun_out/r5_am/bench_tum.json').read().strip().splitlines()[-1]); l=d['latency']; print('tum', {k: round(l[k],3) for k in ('single_pair_align_ms','tracker_frame_from_images_ms','tracker_frame_with_both_score_blocks_ms','tracker_frame_next_frame_staged_ms','lc_batch_align_ms')})
PY
done
timeout -k 10 600 python bench.py --shape eth3d --steps 8 --warmup 4 --no-cpu-baseline > $O/bench_eth.json 2> $O/bench_eth.err || exit 1
python - <<'PY' | tee -a $O/latency.txt
import json; d=json.loads(open('gpurun_out/r5_am/bench_eth.json').read().strip().splitlines()[-1]); l=d['latency']; print('eth3d', round(d['value']), {k: round(l[k],3) for k in ('single_pair_align_ms','lc_batch_align_ms')})
PY
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"
