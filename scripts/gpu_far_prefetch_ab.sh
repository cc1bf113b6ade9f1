#!/bin/bash
# same-lease A/B of the walks' far prefetch (CVO_FAR_PF steps ahead, global_load_lds into a sink); libs built by scripts/build_variant.sh pfN -DCVO_FAR_PF=N
LIBS=${LIBS:-"pf0 pf1 pf2 pf4"}
run() { # shape steps warm lib
  v=$(CVO_BENCH_PHASES=1 CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$4.so timeout -k 10 300 python bench.py --shape $1 --steps $2 --warmup $3 --no-cpu-baseline --no-latency-probe 2>gpurun_out/farpf.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep $1 steps $2 $4: $v | $(grep 'phase us' gpurun_out/farpf.err | sed 's/.*launch): //' | cut -c1-200)"
}
for rep in $(seq 1 ${REPS:-2}); do
  for lib in $LIBS; do run tum 256 16 $lib; done
  for lib in $LIBS; do run tum 20 5 $lib; done
  for lib in $LIBS; do run eth3d 24 4 $lib; done
done
