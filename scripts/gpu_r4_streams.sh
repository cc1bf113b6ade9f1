#!/bin/bash
# steps in flight with the round-4 kernel (one lease)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_streams; mkdir -p $O; hostname > $O/lease.txt
for rep in 1 2; do for s in 6 8 10 12; do for st in "20 5" "256 32"; do set -- $st
r=$(GPU_MAX_HW_QUEUES=$s timeout -k 10 300 python bench.py --streams $s --steps $1 --warmup $2 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
echo "rep $rep streams $s steps $1: $r" | tee -a $O/sweep.txt
done; done; done
