#!/bin/bash
# the shader clock the align kernel sees (s_memtime cycles per s_memrealtime tick, averaged over the pairs of the last launch): one launch alone, 4 and 8 in flight
for st in 1 2 4 8 12; do
  GPU_MAX_HW_QUEUES=16 CVO_BENCH_PHASES=1 CVO_HIP_REPORT_CLOCK=1 timeout -k 10 200 python bench.py --streams $st --steps 128 --warmup 16 --no-cpu-baseline --no-latency-probe 2> gpurun_out/clock_$st.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st:', round(d['value']), 'alignments/s')"
  grep "shader clock" gpurun_out/clock_$st.err | tail -2
done
