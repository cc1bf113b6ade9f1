#!/bin/bash
# PMC passes for the align kernel: one rocprofv3 run per counter group (no trace domains),
# each under its own timeout, progress appended to <outdir>/progress.log.
# usage: [BENCH_ARGS="..."] scripts/pmc_run.sh <outdir> <pass> [<pass> ...]     passes: sq1 sq2 sq3 sq4 sq5 fetch write tcc
OUT=${1:-gpurun_out/pmc}; shift
BENCH_ARGS=${BENCH_ARGS:---steps 3 --warmup 1 --streams 1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
declare -A C
C[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
C[sq2]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
C[sq3]="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F32"
C[sq4]="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_TRANS_F64"
C[sq5]="SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_INSTS_SMEM SQ_INSTS_BRANCH"
C[fetch]="FETCH_SIZE"
C[write]="WRITE_SIZE"
C[tcc]="TCC_HIT_sum TCC_MISS_sum"
for name in "$@"; do
  echo "$(date +%T) start $name: ${C[$name]}  [bench.py $BENCH_ARGS]" >> $OUT/progress.log
  timeout -k 5 200 rocprofv3 --pmc ${C[$name]} --output-format csv -d $OUT/$name -- python3 bench.py $BENCH_ARGS --no-cpu-baseline --no-latency-probe > $OUT/$name.log 2>&1
  echo "$(date +%T) end $name rc=$?" >> $OUT/progress.log
done
cat $OUT/progress.log
