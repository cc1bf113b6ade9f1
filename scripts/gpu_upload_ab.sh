#!/bin/bash
# with_host_upload (bench.py's PCIe-inclusive loop) under the hand-over's knobs, one lease
for rep in 1 2; do for v in "zerocopy_t4 CVO_HIP_UPLOAD_COPY=0" "copy_t4 CVO_HIP_UPLOAD_COPY=1" "zerocopy_t1 CVO_HIP_UPLOAD_THREADS=1" "zerocopy_t8 CVO_HIP_UPLOAD_THREADS=8" "copy_t8 CVO_HIP_UPLOAD_COPY=1 CVO_HIP_UPLOAD_THREADS=8"; do
  label=${v%% *}; envs=${v#* }
  env $envs timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value']), 'upload', round(d['with_host_upload']['value']), round(d['with_host_upload']['value']/d['value'],3), 'score', round(d['with_score_block']['value']), round(d['with_score_block']['value']/d['value'],3))"
done; done
