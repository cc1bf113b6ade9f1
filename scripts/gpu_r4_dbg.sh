#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "far_from or dense_near or workgroup_counts" 2>&1 | tail -30
