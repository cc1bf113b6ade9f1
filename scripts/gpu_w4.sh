#!/bin/bash
# experiment: two 512-thread workgroups per CU (128 VGPRs, 16 waves per CU) with today's LDS layouts, against the default
run() { echo "$1: $(env $2 timeout -k 10 300 python bench.py --pairs $3 --streams $4 --steps $5 --warmup $6 --no-cpu-baseline --no-latency-probe 2>gpurun_out/w4.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), d['config']['iterations_mean'])" || tail -3 gpurun_out/w4.err)" | tee -a gpurun_out/w4.txt; }
: > gpurun_out/w4.txt
W4="CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_w4.so CVO_HIP_WGS_PER_CU=2 CVO_HIP_BLOCK=512"
W41="CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_w4.so"
for rep in 1 2; do
run "default 8 streams, 128 steps" "A=1" 64 8 128 16
run "w4pf2 1/CU 8 streams, 128 steps" "$W41" 64 8 128 16
run "w4pf2 2/CU 16 streams 16 queues, 128 steps" "$W4 GPU_MAX_HW_QUEUES=16" 64 16 128 16
run "w4pf2 2/CU 128 pairs 8 streams, 64 steps" "$W4" 128 8 64 8
done
