#!/bin/bash
# round 5, call X: phase times iteration by iteration, one pair alone (experiment build with timers)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_x; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_ktrace.so timeout -k 10 300 python scripts/r05/probe_iter.py > $O/iter.txt 2>&1; echo rc=$?; cat $O/iter.txt
