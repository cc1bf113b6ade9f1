#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_lc; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_tail_scores.py tests/test_gpu_config3.py -x -q 2>&1 | tail -3 | tee $O/pytest.txt
for rep in 1 2; do for fs in default 1; do
  if [ $fs = default ]; then unset CVO_HIP_FIRST_SCALE; else export CVO_HIP_FIRST_SCALE=$fs; fi
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['latency']; print('first_scale $fs:', round(d['value'],0), {k: round(l[k],3) for k in ('single_pair_align_ms','tracker_frame_from_images_ms','lc_batch_align_ms','lc_batch_score_block_ms')})"
done; done | tee $O/ab.txt
