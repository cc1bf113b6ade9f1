#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in nsf nsm nsboth; do echo "== $l"; CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$l.so python scripts/gpu_r4_fuse_dbg.py 2>&1 | grep -v amdgpu.ids | grep "steps cap 8" | cut -c1-140; done
