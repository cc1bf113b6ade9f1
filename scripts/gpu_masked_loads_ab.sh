#!/bin/bash
# same-lease A/B of the entry-load masking (CVO_MASK_ENT_*) and of non-temporal record loads (CVO_NT_REC_LD); libs built by scripts/build_variant.sh
run() { # shape steps warm lib
  v=$(CVO_HIP_LIB=$PWD/tmp_libs/libcvo_hip_$4.so timeout -k 10 300 python bench.py --shape $1 --steps $2 --warmup $3 --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1))")
  echo "rep $rep $1 steps $2 $4: $v"
}
for rep in 1 2 3; do
  for lib in mnone mall mall_ntld mdef_ntld; do run tum 256 16 $lib; done
  for lib in mnone mall; do run tum 20 5 $lib; done
  for lib in mnone mdef mdef_ntld; do run eth3d 24 4 $lib; done
done
