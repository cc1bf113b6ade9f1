#!/bin/bash
# round 4: depth-proportional list margin: other motion mixes (translation-heavy, rotation-heavy, large), and the fine sweep near the optimum
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_alpha3; mkdir -p $O; hostname > $O/lease.txt
V=("base" "s05a015 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.015" "s10a015 CVO_HIP_SKIN=0.10 CVO_HIP_SKIN_ALPHA=0.015" "s15a01 CVO_HIP_SKIN=0.15 CVO_HIP_SKIN_ALPHA=0.01" "s20a01 CVO_HIP_SKIN=0.20 CVO_HIP_SKIN_ALPHA=0.01")
for m in "0.5,0.06" "3.0,0.01" "4.0,0.06" "0.5,0.01"; do
  echo "motion $m" | tee -a $O/motion.txt
  CVO_BENCH_MOTION=$m bash scripts/gpu_ab_env.sh $O/m.txt 1 "tum 64 16" -- "${V[@]}"; cat $O/m.txt >> $O/motion.txt
done
bash scripts/gpu_ab_env.sh $O/fine.txt 1 "tum 20 5" "tum 256 32" -- "s05a015 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.015" "s05a010 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.010" "s05a0125 CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.0125" "s03a015 CVO_HIP_SKIN=0.03 CVO_HIP_SKIN_ALPHA=0.015" "s03a0175 CVO_HIP_SKIN=0.03 CVO_HIP_SKIN_ALPHA=0.0175" "s08a0125 CVO_HIP_SKIN=0.08 CVO_HIP_SKIN_ALPHA=0.0125" "s08a015 CVO_HIP_SKIN=0.08 CVO_HIP_SKIN_ALPHA=0.015" "s05a015b CVO_HIP_SKIN=0.05 CVO_HIP_SKIN_ALPHA=0.015"
