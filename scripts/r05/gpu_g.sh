#!/bin/bash
# round 5, call G: the whole GPU suite, smoke, the driver's command (complete line incl. the config5 child)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_g; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -3 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -eq 124 ]; then exit 1; fi
timeout -k 10 120 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_g/bench_driver.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "frac", round(d["roofline"]["frac"], 3), "score", round(d["with_score_block"]["value"] / d["value"], 3), "upload", round(d["with_host_upload"]["value"] / d["value"], 3),
      "parity", d["parity"]["max_rot_err_rad"], d["parity"]["max_trans_err_m"], "cpu", round(d["cpu_baseline"]["value"], 1))
print("latency", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["latency"].items() if k != "note"})
print("config5", json.dumps(d.get("config5"))[:1500])
PY
