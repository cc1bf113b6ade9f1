#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_h; mkdir -p $O
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python scripts/r05/probe_stage.py 2>&1 | tee $O/probe_stage.txt
