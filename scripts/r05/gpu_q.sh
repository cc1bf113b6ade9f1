#!/bin/bash
# round 5, call Q: the epilogue's timeline (experiment build with the epilogue sub-timers, CVO_KTRACE_EPI=2)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_q; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_${1:-epi2}.so timeout -k 10 300 python scripts/r05/probe_epi2.py > $O/epi2_${1:-epi2}.txt 2>&1; echo rc=$?; cat $O/epi2_${1:-epi2}.txt
