#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_fuse_eth; mkdir -p $O; hostname > $O/lease.txt
timeout -k 10 600 python -m pytest tests/test_gpu_config5.py tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1; echo "parity rc=$? $(tail -1 $O/pytest.txt)"
bash scripts/gpu_ab_env.sh $O/ab.txt 2 "eth3d 24 4" -- "pass CVO_HIP_FUSE_REFINE=0" "fused CVO_HIP_FUSE_REFINE=1"
