#!/bin/bash
# A/B of environment knobs on ONE lease, interleaved.  usage: gpu_ab_env.sh OUT REPS "shape steps warmup" ... -- "label ENV=val ENV=val" ...
OUT=$1; REPS=$2; shift 2
CFGS=(); while [ "$1" != "--" ]; do CFGS+=("$1"); shift; done; shift
: > $OUT
VARS=("$@")
for rep in $(seq 1 $REPS); do for cfg in "${CFGS[@]}"; do for v in "${VARS[@]}"; do
  read -r shape steps warm <<< "$cfg"
  label=${v%% *}; envs=${v#* }; [ "$envs" = "$label" ] && envs=""
  r=$(env $envs CVO_BENCH_PHASES=1 timeout -k 10 300 python bench.py --shape $shape --steps $steps --warmup $warm --no-cpu-baseline --no-latency-probe 2> gpurun_out/ab_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['kernel_ms'],2))") || { echo "run failed: $label $cfg" | tee -a $OUT; tail -3 gpurun_out/ab_err.txt | tee -a $OUT; exit 1; }
  echo "rep $rep $shape steps $steps [$label]: $r | $(grep -o 'phase us.*' gpurun_out/ab_err.txt | sed -e 's/phase us.iteration under load (workgroup 0 of every pair of the last launch): //' | cut -c1-330)" | tee -a $OUT
done; done; done
