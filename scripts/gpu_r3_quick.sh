#!/bin/bash
# quick check of a kernel change on one lease: the parity tests that exercise the align kernel, then phases + short and long bench, each A/B against CVO_HIP_KEEP_FLOOR=1 (colour-blind lists)
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_noise_envelope.py -m gpu -q -x > gpurun_out/r3_q_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_q_tests.log
if [ $rc -ne 0 ]; then exit $rc; fi
for kf in 1 0.5 1 0.5; do
  for cfg in "20 5" "256 32"; do set -- $cfg
    v=$(CVO_HIP_KEEP_FLOOR=$kf CVO_BENCH_PHASES=1 timeout -k 10 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-latency-probe 2> gpurun_out/r3_q_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2))")
    echo "keep_floor $kf steps $1: $v  $(grep -o 'phase us.*' gpurun_out/r3_q_err.txt | cut -c1-400)"
  done
done
