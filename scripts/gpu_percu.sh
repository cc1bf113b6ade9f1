#!/bin/bash
# experiment: two 256-thread workgroups per CU (12-byte y planes + small cull tile, x_i from L2) against the default
run() { echo "$1: $(env $2 timeout -k 10 300 python bench.py --pairs $3 --streams $4 --steps $5 --warmup 16 --no-cpu-baseline --no-latency-probe 2>gpurun_out/percu.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2))" || tail -3 gpurun_out/percu.err)"; }
run "default 64 pairs x 8 streams" "A=1" 64 8 128
run "default 128 pairs x 8 streams" "A=1" 128 8 64
run "2 per CU, 64 pairs x 8 streams" "CVO_HIP_WGS_PER_CU=2 CVO_HIP_Y_MODE=2 CVO_HIP_TILE=1024" 64 8 128
run "2 per CU, 128 pairs x 8 streams" "CVO_HIP_WGS_PER_CU=2 CVO_HIP_Y_MODE=2 CVO_HIP_TILE=1024" 128 8 64
run "2 per CU, 128 pairs x 8 streams, tile 512" "CVO_HIP_WGS_PER_CU=2 CVO_HIP_Y_MODE=2 CVO_HIP_TILE=512" 128 8 64
