#!/bin/bash
# Adoption A/B on ONE lease: --no-adoption and --adoption interleaved, five times, for the driver's command (20 steps, 5 warm-up) and
# the 256-step default.  bench.py's flags decide the mode (the CVO_HIP_ADOPT environment knob is overridden by bench.py).
OUT=${1:-gpurun_out/adopt_ab.txt}; : > $OUT
for rep in 1 2 3 4 5; do for cfg in "20 5" "256 32"; do for mode in --no-adoption --adoption; do
  set -- $cfg
  v=$(timeout -k 10 200 python bench.py --steps $1 --warmup $2 $mode --no-cpu-baseline --no-latency-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms'],2), d['config']['adoption'])") || { echo "run failed ($mode $cfg)" | tee -a $OUT; exit 1; }
  echo "rep $rep steps $1 $mode: $v" | tee -a $OUT
done; done; done
python - "$OUT" <<'PY'
import re, sys, statistics as st
rows = [re.match(r"rep (\d+) steps (\d+) (--\S+): (\d+)", l) for l in open(sys.argv[1])]
rows = [(int(m.group(2)), m.group(3), int(m.group(4))) for m in rows if m]
with open(sys.argv[1], "a") as f:
    for steps in (20, 256):
        off = [v for s, m, v in rows if s == steps and m == "--no-adoption"]; on = [v for s, m, v in rows if s == steps and m == "--adoption"]
        if off and on:
            line = f"steps {steps}: off median {st.median(off):.0f} (min {min(off)}, max {max(off)}), on median {st.median(on):.0f} (min {min(on)}, max {max(on)}), gain {100 * (st.median(on) / st.median(off) - 1):+.1f} %"
            print(line); f.write(line + "\n")
PY
