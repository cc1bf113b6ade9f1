#!/bin/bash
# round 5, call I: hand-over from registered memory, staged frames, gather stream: their tests; then the driver's command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_i; mkdir -p $O; date -u +%FT%TZ > $O/lease.txt
timeout -k 10 900 python -m pytest tests/test_gpu_handover.py tests/test_gpu_pcd.py tests/test_gpu_multi.py tests/test_gpu_replay.py -x -q > $O/pytest.txt 2>&1; rc=$?; echo "pytest rc=$rc $(tail -3 $O/pytest.txt | tr '\n' ' ')"; if [ $rc -eq 124 ]; then exit 1; fi
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "driver rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_i/bench_driver.json").read().strip().splitlines()[-1])
print("value", round(d["value"], 1), "score", round(d["with_score_block"]["value"] / d["value"], 3), "upload", round(d["with_host_upload"]["value"] / d["value"], 3),
      "registered", d["with_host_upload"]["from_registered_memory"] and round(d["with_host_upload"]["from_registered_memory"]["value"] / d["value"], 3), "parity", d["parity"]["max_rot_err_rad"], d["parity"]["max_trans_err_m"])
print("latency", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["latency"].items() if k.startswith("tracker") or k.startswith("single")})
c5 = d.get("config5") or {}
print("config5", c5.get("value"), c5.get("roofline", {}).get("frac"), c5.get("parity", {}).get("max_rot_err_rad"), c5.get("child_seconds"), c5.get("error"))
PY
tail -3 $O/bench_driver.err
