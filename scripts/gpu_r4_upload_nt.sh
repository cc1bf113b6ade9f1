#!/bin/bash
# hand-over copies into the pinned ring with non-temporal stores (CVO_HIP_UPLOAD_NT=1) against memcpy (0): the with_host_upload loop of the driver's command and of a 64-step run, interleaved on one lease
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_upload_nt; mkdir -p $O; hostname > $O/lease.txt; : > $O/ab.txt
for rep in 1 2; do for steps in "20 5" "64 8"; do for nt in 0 1; do
  read -r K W <<< "$steps"
  r=$(CVO_HIP_UPLOAD_NT=$nt timeout -k 10 300 python bench.py --steps $K --warmup $W --no-cpu-baseline 2> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['with_host_upload']; print(round(d['value'],1), round(u['value'],1), round(u['value']/d['value'],4), u.get('host_ms_per_step'))") || { echo failed; tail -3 $O/err.txt; exit 1; }
  echo "rep $rep steps $K nt=$nt: value, with_host_upload, ratio, host ms/step: $r" | tee -a $O/ab.txt
done; done; done
