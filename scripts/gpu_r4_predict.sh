#!/bin/bash
# round 4: candidate lists built / filtered around extrapolated positions (DevParams::predict): parity with it on, then the fraction of the allowance, one lease
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_predict; mkdir -p $O; hostname > $O/lease.txt
CVO_HIP_PREDICT=0.7 timeout -k 10 400 python -m pytest tests/test_gpu_config3.py tests/test_gpu_adoption.py tests/test_gpu_tail_scores.py tests/test_gpu_parity.py tests/test_gpu_config5.py -x -q > $O/pytest_predict.txt 2>&1; echo "parity with predict 0.7 rc=$? $(tail -1 $O/pytest_predict.txt)"
for p in 0 0.7; do CVO_HIP_PREDICT=$p CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-latency-probe 2>&1 >/dev/null | grep "culls by iteration" | sed -e 's/ 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0 0;/;/'; done
bash scripts/gpu_ab_env.sh $O/sweep.txt 1 "tum 20 5" "tum 256 32" -- "base" "p5 CVO_HIP_PREDICT=0.5" "p7 CVO_HIP_PREDICT=0.7" "p9 CVO_HIP_PREDICT=0.9" "p7s3 CVO_HIP_PREDICT=0.7 CVO_HIP_PREDICT_STEPS=3" "p9s20 CVO_HIP_PREDICT=0.9 CVO_HIP_PREDICT_STEPS=20" "base2"
