#!/bin/bash
# round 5, call AO: config 5 with the final kernel: workgroups per pair x workgroups per launch x launches side by side
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_ao; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt; : > $O/sweep.txt
for cfg in "4 64 4" "8 64 4" "8 128 2" "2 64 4" "4 128 2" "6 48 5" "4 64 4"; do set -- $cfg
  r=$(timeout -k 10 300 python bench.py --shape eth3d --steps 12 --warmup 4 --workgroups $1 --max-workgroups $2 --streams $3 --no-cpu-baseline --no-latency-probe --parity-only 2> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), d['parity']['max_rot_err_rad'])") || { echo "failed: $cfg"; tail -3 $O/err.txt; continue; }
  echo "workgroups per pair $1, per launch $2, launches $3: $r" | tee -a $O/sweep.txt
done
