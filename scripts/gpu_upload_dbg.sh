#!/bin/bash
for rep in 1 2; do for v in "copy CVO_HIP_UPLOAD_COPY=1" "hostonly CVO_HIP_UPLOAD_COPY=1 CVO_HIP_UPLOAD_DEBUG=1" "packonly CVO_HIP_UPLOAD_COPY=1 CVO_HIP_UPLOAD_DEBUG=2"; do
  label=${v%% *}; envs=${v#* }
  env $envs timeout -k 10 300 python bench.py --steps 40 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['value']), 'upload', round(d['with_host_upload']['value']), round(d['with_host_upload']['value']/d['value'],3))"
done; done
