#!/bin/bash
# instruction-class counters of the align kernel (for the cost-weighted VALU roofline) + the same SQ counters on the issue-rate microbenchmark
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_classes; rm -rf $O; mkdir -p $O
bash scripts/pmc_run.sh $O sq4 sq5
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/micro -- scripts/micro/valu_rate > $O/micro.txt 2> $O/micro.err; echo "micro rc=$?"
find $O -name "*counter_collection.csv" | head; du -sh $O
