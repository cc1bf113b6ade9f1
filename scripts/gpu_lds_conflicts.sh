#!/bin/bash
# where the LDS bank conflicts come from: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the align kernel (one launch alone) and the bench's throughput under layout knobs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/lds_r03; rm -rf $O; mkdir -p $O
for v in "default X=1" "no_table CVO_HIP_NO_TABLE=1" "y_planes CVO_HIP_Y_MODE=2"; do
  label=${v%% *}; envs=${v#* }
  env $envs BENCH_ARGS="--steps 3 --warmup 1 --streams 1 --no-adoption" bash scripts/pmc_run.sh $O/$label sq3 > /dev/null 2>&1
  python scripts/pmc_summarize.py $O/$label $O/$label.json > /dev/null 2>&1
  r=$(python -c "import json; d=json.load(open('$O/$label.json'))['per_launch']; print(round(d['SQ_LDS_BANK_CONFLICT']/d['SQ_LDS_IDX_ACTIVE'],3), int(d['SQ_LDS_IDX_ACTIVE']), int(d['SQ_LDS_BANK_CONFLICT']))")
  t=$(env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']))")
  echo "$label: conflict/active, active, conflict cycles = $r; alignments/s $t"
done
