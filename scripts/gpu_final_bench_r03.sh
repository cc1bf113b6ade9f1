#!/bin/bash
# the complete bench lines at HEAD, no profiler: the driver's command, the default command, the config-5 shape
cd $GRAFT_REPO_ROOT; O=gpurun_out/final_r03; mkdir -p $O
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc=$?"
timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
timeout -k 10 600 python bench.py --shape eth3d > $O/bench_eth3d.json 2> $O/bench_eth3d.err; echo "eth3d rc=$?"
for f in driver default eth3d; do python -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
print('$f', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'cpu', round(d['cpu_baseline']['value'],1), 'x', round(d['speedup_vs_cpu_baseline']), 'parity', d['parity']['max_rot_err_rad'], d['parity']['max_trans_err_m'], d['parity']['iterations_equal'], 'score', d['with_score_block'] and round(d['with_score_block']['value']), 'upload', d['with_host_upload'] and round(d['with_host_upload']['value']))"; done
