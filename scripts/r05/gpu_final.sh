#!/bin/bash
# round 5, final call: the full GPU suite, smoke, and the three bench lines kept under profiles/ (driver's command first, as the driver runs it)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_final; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -2 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -ne 0 ]; then tail -30 $O/pytest.txt; exit 1; fi
timeout -k 10 300 python __graft_entry__.py --smoke > $O/smoke.txt 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.txt | cut -c1-200)"
t0=$(date +%s); timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err || exit 1; echo "driver's command: $(( $(date +%s) - t0 )) s wall"
timeout -k 10 900 python bench.py --no-config5 > $O/bench_default.json 2> $O/bench_default.err || exit 1
timeout -k 10 900 python bench.py --shape eth3d > $O/bench_eth3d.json 2> $O/bench_eth3d.err || exit 1
python - <<'PY'
import json
for n in ("driver", "default", "eth3d"):
    d = json.loads(open(f"gpurun_out/r5_final/bench_{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"]), "frac", round(d["roofline"]["frac"], 3), "parity", d["parity"]["max_rot_err_rad"], d["parity"]["max_trans_err_m"], "cpu", round(d.get("cpu_baseline", {}).get("value", 0), 1),
          "config5", (d.get("config5") or {}).get("value"), "upload", (d.get("with_host_upload") or {}).get("fraction_of_value"), "scores", (d.get("with_score_block") or {}).get("fraction_of_value"), "latency", {k: round(v, 3) for k, v in (d.get("latency") or {}).items() if k.endswith("_ms") and ("single_pair_align" in k or "tracker_frame" in k or "lc_batch_align" in k)})
PY
