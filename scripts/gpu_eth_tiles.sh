#!/bin/bash
# BASELINE config 5 "LDS tile-size sweep": columns of the cull staged through an LDS tile of T points (y in HBM/L2: layout 0) for
# T = 512 ... 4096, against the plane layout (y resident in LDS as three float planes, one pass, no tile)
: > gpurun_out/eth_tiles.txt
run() { v=$(env $2 CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --shape eth3d --steps 16 --warmup 4 --no-cpu-baseline --no-latency-probe 2>gpurun_out/eth_tiles.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],1))"); ph=$(grep "phase us" gpurun_out/eth_tiles.err | sed 's/.*launch): //' | python -c "import sys,ast; d=ast.literal_eval(sys.stdin.read()); print('cull', d['lists_cull'], 'cand', d['candidates'], 'ls', d['linesearch'])"); echo "$1: $v $ph" | tee -a gpurun_out/eth_tiles.txt; }
run "planes (layout 2, default)" "A=1"
for T in 4096 2048 1024 512; do run "tile $T (layout 0)" "CVO_HIP_Y_MODE=0 CVO_HIP_TILE=$T"; done
