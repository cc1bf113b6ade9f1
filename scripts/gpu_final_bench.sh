#!/bin/bash
# the three bench lines committed under profiles/ (driver-style, default, BASELINE config 5 shape), each the complete default command
mkdir -p gpurun_out/final
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/driver.json 2> gpurun_out/final/driver.err && echo "driver ok" &&
timeout -k 10 500 python bench.py > gpurun_out/final/default.json 2> gpurun_out/final/default.err && echo "default ok" &&
timeout -k 10 600 python bench.py --shape eth3d > gpurun_out/final/eth3d.json 2> gpurun_out/final/eth3d.err && echo "eth3d ok"
python - <<'PY'
import json
for n in ("driver", "default", "eth3d"):
    try:
        d = json.loads(open(f"gpurun_out/final/{n}.json").read().strip().splitlines()[-1])
        r = d["roofline"]; c = d.get("cpu_baseline", {}); p = d.get("parity", {})
        print(n, round(d["value"]), "frac", round(r["frac"], 3), "of measured", round(r["frac_of_measured_issue_rate"], 3), "old", round(r["frac_of_one_wave_issue_rate"], 3), "kernel_ms", round(r["kernel_ms"], 2),
              "cpu", round(c.get("value", 0), 1), "parity", p.get("max_rot_err_rad"), p.get("max_trans_err_m"), p.get("iterations_equal"))
    except Exception as e:
        print(n, "failed", e)
PY
