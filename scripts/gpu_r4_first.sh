#!/bin/bash
# round 4, first call: the GPU suite at HEAD, the N > 1 rehearsal through the bare invocation, the driver's command (this lease's baseline)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_first; mkdir -p $O
hostname > $O/lease.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.txt)"
bash scripts/gpu_n2.sh 2>&1 | tee $O/n2.txt
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc=$?"
python -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('driver', round(d['value'],1), 'frac', round(d['roofline']['frac'],3), 'score', round(d['with_score_block']['value']), 'upload', round(d['with_host_upload']['value']), 'distinct', d['distinct_pairs'] and round(d['distinct_pairs']['value']), 'lane-instr/nz', d['work']['valu_lane_instructions_per_nonzero'], 'tracker ms', d['latency']['tracker_frame_from_images_ms'])"
