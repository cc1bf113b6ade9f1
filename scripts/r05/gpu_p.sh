#!/bin/bash
# round 5, call P: adoption only after the device's queue has been dry for a while (CVO_HIP_ADOPT_DWELL_US): 0 / 100 / 300 / 1000 us
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_p; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 600 python -m pytest tests/test_gpu_adoption.py tests/test_gpu_config3.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then exit 1; fi
CVO_HIP_ADOPT_DWELL_US=300 timeout -k 10 600 python -m pytest tests/test_gpu_adoption.py tests/test_gpu_config3.py -x -q > $O/pytest_dwell.txt 2>&1; rc=$?; echo "pytest (dwell 300) rc=$rc $(tail -2 $O/pytest_dwell.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then exit 1; fi
bash scripts/gpu_ab_env.sh $O/ab.txt 3 "tum 20 5" "tum 256 32" -- "dwell0" "dwell100 CVO_HIP_ADOPT_DWELL_US=100" "dwell300 CVO_HIP_ADOPT_DWELL_US=300" "dwell1000 CVO_HIP_ADOPT_DWELL_US=1000" | cut -c1-120
