#!/bin/bash
# the GPU suite, then the driver's command and the default run with the library's defaults
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_suite; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.txt)"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('driver', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'score', round(d['with_score_block']['value']), 'upload', round(d['with_host_upload']['value']), 'distinct', d['distinct_pairs'] and round(d['distinct_pairs']['value']), 'parity', d['parity']['max_rot_err_rad'], d['parity']['max_trans_err_m'], 'tracker ms', d['latency']['tracker_frame_from_images_ms'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-latency-probe > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default', round(d['value'],1))"
